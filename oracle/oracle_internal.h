/* oracle_internal.h — types shared between the oracle's translation units.  TEST INFRASTRUCTURE. */
#ifndef ORACLE_INTERNAL_H
#define ORACLE_INTERNAL_H
#include "oracle.h"
#include "oracle_math.h"

typedef struct { double lat, lon, elev; } ocoords; /* utils/mod.rs:8-13 */

/* SerializableObject after Altitude::abs (object/mod.rs:164-183) */
typedef struct {
  int kind;
  double lat, lon, elev;
  double r1, r2, height, width;
  double color[4];
  const uint8_t* tex;
  uint32_t tex_w, tex_h;
} oracle_object;

typedef struct {
  double prop;
  ovec3 normal;
  double color[4];
} oracle_collision;

int oracle_object_collision(const oracle_object* o, const atmrt_earth_model_t* m, ocoords p1, ocoords p2,
                            oracle_collision* out /* >= 4 */);
int oracle_object_is_close(const oracle_object* o, const atmrt_earth_model_t* m, double sim_step, double lat, double lon);
#endif

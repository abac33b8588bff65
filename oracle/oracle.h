/* oracle.h — CPU restatement of atm-raytracer's per-pixel ray-marching path.
 *
 * TEST INFRASTRUCTURE.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library; the product (atm-raytracer_amd/) never does.
 *
 * PARITY UNPINNED: the reference ships no test, fixture or golden vector for this path
 * (SURVEY.md §4, §8c), it cannot be built here (no Rust toolchain), and two of its numerical
 * kernels live in crates whose sources are absent: `atm-refraction` 0.6 (ray ODE, atmosphere,
 * refractive index) and `dted` 0.2 (DTED parsing, bilinear sampling).  Everything that IS in the
 * reference repository is restated line by line (citations on each function, paths relative to
 * /root/reference); the two absent crates are restated from their published models, with every
 * modelling choice listed in DESIGN.md §"Unpinned choices".  Known-answer tests
 * (tests/test_oracle_*.py) pin the restatement to closed-form physics and geodesy instead.
 */
#ifndef ORACLE_H
#define ORACLE_H

#include "../include/atmrt.h" /* POD parameter / result structs shared with the C ABI */

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double x, y, z; } ovec3;

/* ---- atmosphere + refractive index (crate atm-refraction, absent) ------------------------- */
typedef struct { /* the atmosphere compiled into segments: T(h) = tb + c1 dh + c2 dh^2 + c3 dh^3, dh = h - hb.  Any number of
                    segments (Vec in the reference, params.rs:453-454): the arrays are allocated by oracle_atm_compile */
  int n;
  double* hb;    /* reference altitude of the segment */
  double* tb;    /* temperature at hb */
  double* pb;    /* pressure at hb */
  double* lapse; /* c1 */
  double* from;  /* segment k>=1 applies for h >= from[k] */
  double* expo;  /* linear: lapse != 0 ? -g0 M/(R lapse) : -g0 M/(R tb);  cubic: -g0 M/R */
  double* c2;
  double* c3;
  int* cubic;    /* 1: a knot interval of a Spline temperature function */
  double k_refr; /* (n-1) = k_refr * (p/T) / Z */
} oracle_env_atm;

void oracle_atmosphere_us76(atmrt_atmosphere_t* a);
int oracle_atm_compile(const atmrt_atmosphere_t* def, double wavelength, oracle_env_atm* out); /* release with oracle_atm_free, also after a failure */
void oracle_atm_free(oracle_env_atm* a);
double oracle_atm_temperature(const oracle_env_atm* a, double h);
double oracle_atm_pressure(const oracle_env_atm* a, double h);
double oracle_n(const oracle_env_atm* a, double h);
double oracle_dn(const oracle_env_atm* a, double h);

/* ---- earth model / geodesy (src/utils/earth_model) ---------------------------------------- */
typedef struct {
  int kind;        /* 0 AzEq, 1 FlDs, 2 Spherical, 3 Ellipsoid */
  double radius;   /* Spherical */
  ovec3 pos, dir;  /* Spherical: unit position / tangent; AzEq: cartesian pos / dir_v */
  double start_lat, start_lon, dir_deg; /* FlDs */
  double b, f, red_lat, lon, az1, alfa, sig1, cap_a, cap_b, cap_c; /* Ellipsoid */
} oracle_dircalc;

void oracle_world_directions(const atmrt_earth_model_t* m, double lat, double lon, ovec3* n, ovec3* e, ovec3* up);
ovec3 oracle_as_cartesian(const atmrt_earth_model_t* m, double lat, double lon, double elev);
/* returns 1 and sets *radius for EarthShape::Spherical, 0 for EarthShape::Flat */
int oracle_to_shape(const atmrt_earth_model_t* m, double* radius);
void oracle_dircalc_new(const atmrt_earth_model_t* m, double lat, double lon, double dir_deg, oracle_dircalc* out);
void oracle_coords_at_dist(const oracle_dircalc* c, double dist, double* lat, double* lon);

/* ---- ray stepper (crate atm-refraction, absent) ------------------------------------------- */
typedef struct { double x, h, dh; } oracle_ray_state;
typedef struct {
  const oracle_env_atm* atm;
  int spherical, straight;
  double radius, step;
  double x, a, b;   /* flat: a = h, b = dh/dx.  spherical: a = r, b = dr/dphi */
  double h0, ang;   /* straight rays: closed form from the start */
} oracle_stepper;
void oracle_stepper_init(oracle_stepper* s, const oracle_env_atm* atm, int spherical, double radius, double h0,
                         double ang_rad, int straight, double step);
oracle_ray_state oracle_stepper_next(oracle_stepper* s);

/* ---- terrain (src/terrain + crate dted, absent) ------------------------------------------- */
typedef struct {
  int lat0, lon0, n_lat, n_lon;
  int16_t* posts; /* [n_lat][n_lon], south->north rows, west->east posts */
} oracle_tile;
typedef struct {
  oracle_tile* tiles;
  int n_tiles, cap;
} oracle_terrain;

oracle_terrain* oracle_terrain_new(void);
void oracle_terrain_free(oracle_terrain* t);
int oracle_terrain_add_tile(oracle_terrain* t, int lat0, int lon0, int n_lat, int n_lon, const int16_t* posts);
/* Terrain::from_folder: returns number of files, <0 on error */
int oracle_terrain_load_dir(oracle_terrain* t, const char* path);
/* Terrain::get_elev: returns 1 and writes *elev, or 0 for None */
int oracle_terrain_get_elev(const oracle_terrain* t, double lat, double lon, double* elev);
/* DTED level-n writer used to build synthetic fixtures (round trip with the reader) */
int oracle_dted_write(const char* path, int lat0, int lon0, int n_lat, int n_lon, const int16_t* posts);
int oracle_dted_read(const char* path, int* lat0, int* lon0, int* n_lat, int* n_lon, int16_t** posts);

/* find_normal, utils.rs:15-40 */
ovec3 oracle_find_normal(const atmrt_earth_model_t* model, double lat, double lon, const oracle_terrain* terrain);

/* ---- generators (src/generator/generators) ------------------------------------------------ */
int oracle_generate(const atmrt_params_t* params, const atmrt_atmosphere_t* atm, const oracle_terrain* terrain,
                    const atmrt_object_t* objects, size_t n_objects, int n_threads, atmrt_result_t* out);
void oracle_result_free(atmrt_result_t* r);
/* checker economy for full-size frames: only rows y % stride == phase are computed by the Fast and Rectilinear generators
 * (the others are left without trace points); stride 1 restores the whole frame */
void oracle_set_row_filter(int stride, int phase);

/* ---- renderer compositing + colouring (src/renderer/mod.rs:367-414, src/coloring) ---- */
int oracle_coloring_from_conf(const atmrt_params_t* p, int32_t kind, double water_level, double ambient_light, double light_zenith_angle,
                              double light_dir, int32_t palette, int32_t has_fog, double fog_distance, atmrt_coloring_t* out);
int oracle_draw_image(const atmrt_result_t* r, const atmrt_coloring_t* c, uint8_t* rgb);

/* ---- harnesses ---------------------------------------------------------------------------- */
int oracle_ray_paths(const atmrt_params_t* params, const atmrt_atmosphere_t* atm, double h0, size_t n_angles,
                     const double* angles_deg, int straight, double step, size_t n_steps, double* x, double* h);
const char* oracle_flavour(void);

#ifdef __cplusplus
}
#endif
#endif

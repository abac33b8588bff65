/* oracle_math.h — elementary functions used by the CPU oracle.  TEST INFRASTRUCTURE ONLY.
 *
 * Two build flavours of the same oracle sources (see Makefile):
 *   liboracle_libm.so  (-DORACLE_LIBM): glibc libm — what the Rust reference itself calls on Linux
 *                      (f64::sin etc. lower to the platform libm).  Shares no numerics with the product.
 *   liboracle_det.so   (default): the deterministic functions of atm-raytracer_amd/csrc/detmath.h,
 *                      so CPU and GPU execute one IEEE-754 operation sequence and results can be
 *                      compared BIT-EXACTLY (hit/miss, step indices, every f64 field).
 * tests/ checks det against libm (<= 2 ulp per function, <= 1e-9 relative per output field), and
 * the GPU against det (bit-exact) and against libm (north-star tolerance 1e-4 relative).
 */
#ifndef ORACLE_MATH_H
#define ORACLE_MATH_H

#ifdef ORACLE_LIBM
#include <math.h>
#define om_sin sin
#define om_cos cos
#define om_tan tan
#define om_asin asin
#define om_atan atan
#define om_atan2 atan2
#define om_exp exp
#define om_log log
#define om_pow pow
#define om_sqrt sqrt
#define om_floor floor
#define om_fabs fabs
#define om_fma fma
#define OM_PI 3.14159265358979323846
#define OM_FLAVOUR "libm"
#else
#include "../atm-raytracer_amd/csrc/detmath.h"
#define om_sin dm_sin
#define om_cos dm_cos
#define om_tan dm_tan
#define om_asin dm_asin
#define om_atan dm_atan
#define om_atan2 dm_atan2
#define om_exp dm_exp
#define om_log dm_log
#define om_pow dm_pow
#define om_sqrt dm_sqrt
#define om_floor dm_floor
#define om_fabs dm_fabs
#define om_fma(a, b, c) __builtin_fma((a), (b), (c))
#define OM_PI DM_PI
#define OM_FLAVOUR "det"
#endif

/* Rust: f64::to_radians = self * (PI / 180.0); f64::to_degrees = self * (180.0 / PI). */
static inline double om_to_radians(double d) { return d * (OM_PI / 180.0); }
static inline double om_to_degrees(double r) { return r * (180.0 / OM_PI); }

#endif

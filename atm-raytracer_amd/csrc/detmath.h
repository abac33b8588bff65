/* detmath.h — deterministic f64 elementary functions, identical on host and gfx950.
 *
 * Why this exists: the reference (Rust) calls the platform libm for sin/cos/asin/atan2/exp/ln
 * (e.g. directional_calc.rs:72-85, rectilinear.rs:78-100).  glibc's libm and AMD's OCML round
 * those functions differently in the last bit, and a last-bit difference in a terrain or ray
 * elevation can flip the strict sign test of the tracer (utils.rs:222).  To make hit/miss and
 * step indices bit-identical between the CPU checker and the HIP kernels, both sides evaluate
 * the SAME sequence of IEEE-754 binary64 operations (+ - * / sqrt and fma are correctly rounded
 * on x86-64 and on gfx950; implicit contraction is disabled with -ffp-contract=off on both
 * compilers, so an FMA happens exactly where DM_FMA is written).
 *
 * The algorithms are the classical published ones: Cody–Waite reduction + minimax polynomial
 * kernels in the style of Sun's fdlibm (K.C. Ng 1993) for sin cos tan atan atan2 asin; table-driven
 * exp and log after P.T.P. Tang (1989/1990) with the tables of detmath_tables.h.
 * tests/test_detmath.py bounds every function at <= 1 ulp (tan 2, pow 4) against glibc and mpmath
 * over the argument ranges the tracer uses.  dm_div / dm_div_r / dm_sqrt_inrange are GPU instruction
 * sequences that return the bits of the IEEE operation for in-range operands (host: the operation).
 *
 * Plain C99 / C++: include with DM_FN predefined to add __host__ __device__ in HIP code.
 */
#ifndef ATMRT_DETMATH_H
#define ATMRT_DETMATH_H

#include <stdint.h>
#include "detmath_tables.h"

#ifndef DM_FN
#define DM_FN static inline
#endif

#define DM_PI 3.141592653589793
#define DM_RAD_PER_DEG (DM_PI / 180.0) /* Rust f64::to_radians: self * (PI / 180.0) */
#define DM_DEG_PER_RAD (180.0 / DM_PI) /* Rust f64::to_degrees: self * (180.0 / PI) */

DM_FN uint64_t dm_bits(double x) {
  uint64_t u;
  __builtin_memcpy(&u, &x, 8);
  return u;
}
DM_FN double dm_from_bits(uint64_t u) {
  double x;
  __builtin_memcpy(&x, &u, 8);
  return x;
}
DM_FN uint32_t dm_hi(double x) { return (uint32_t)(dm_bits(x) >> 32); }
DM_FN double dm_fabs(double x) { return dm_from_bits(dm_bits(x) & 0x7fffffffffffffffULL); }
DM_FN double dm_copysign(double m, double s) {
  return dm_from_bits((dm_bits(m) & 0x7fffffffffffffffULL) | (dm_bits(s) & 0x8000000000000000ULL));
}
DM_FN int dm_isnan(double x) { return x != x; }
DM_FN int dm_isinf(double x) { return (dm_bits(x) & 0x7fffffffffffffffULL) == 0x7ff0000000000000ULL; }
DM_FN double dm_inf(void) { return dm_from_bits(0x7ff0000000000000ULL); }
/* exact operations provided by the hardware on both sides */
DM_FN double dm_floor(double x) { return __builtin_floor(x); }
DM_FN double dm_rint(double x) { return __builtin_rint(x); } /* ties-to-even */
DM_FN double dm_sqrt(double x) { return __builtin_sqrt(x); } /* correctly rounded (verified on gfx950 by tests) */
DM_FN double dm_to_radians(double deg) { return deg * DM_RAD_PER_DEG; }
DM_FN double dm_to_degrees(double rad) { return rad * DM_DEG_PER_RAD; }

/* Fused multiply-add (one rounding) and exact scaling by 2^e.  detmath's own polynomials and reductions use them on both
 * sides; the formulas restated from the reference never do (Rust does not contract a*b+c). */
#define DM_FMA(a, b, c) __builtin_fma((a), (b), (c))
/* The same FMA with the PLACE of its constants spelled out for gfx950.  A VALU instruction of gfx9 reads at most one scalar
 * operand and no 64-bit literal, and hipcc's choice for `fma(r, K1, K0)` / `fma(r, q, K)` is the two-address v_fmac with the constant
 * addend copied into a VGPR pair first: 15 v_mov per RK4 stage in the marching kernels (profiles/r03: 228 "other" lane-instructions
 * per ray-step).  Written as VOP3 v_fma_f64 with the constant in an SGPR pair (two s_mov: scalar unit, not a VALU slot) the copies
 * disappear.  Same operation, same rounding: the value is DM_FMA's on both sides.
 *   DM_FMA_VVS(a, b, K)      a * b + K      K: scalar constant
 *   DM_FMA_VSV(a, K, v)      a * K + v      K: scalar constant, v: a value that lives in a VGPR (possibly itself a constant)
 *   DM_FNMA_VVV(a, b, c)     c - a * b      the residual of a Newton step: the negation is a source modifier, not a v_xor + v_mov */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(DM_PLAIN_FMA)
static __device__ __forceinline__ double dm_fma_vvs_(double a, double b, double k) {
  double d;
  __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "s"(k));
  return d;
}
static __device__ __forceinline__ double dm_fma_vsv_(double a, double k, double v) {
  double d;
  __asm__("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(a), "s"(k), "v"(v));
  return d;
}
static __device__ __forceinline__ double dm_fnma_vvv_(double a, double b, double c) {
  double d;
  __asm__("v_fma_f64 %0, -%1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c));
  return d;
}
#define DM_FMA_VVS(a, b, k) dm_fma_vvs_((a), (b), (k))
#define DM_FMA_VSV(a, k, v) dm_fma_vsv_((a), (k), (v))
#define DM_FNMA_VVV(a, b, c) dm_fnma_vvv_((a), (b), (c))
#else
#define DM_FMA_VVS(a, b, k) DM_FMA((a), (b), (k))
#define DM_FMA_VSV(a, k, v) DM_FMA((a), (k), (v))
#define DM_FNMA_VVV(a, b, c) DM_FMA(-(a), (b), (c))
#endif
#if defined(__HIP_DEVICE_COMPILE__)
#define DM_SCALBN(y, e) __builtin_ldexp((y), (e))
#else
#define DM_SCALBN(y, e) ((y) * dm_from_bits((uint64_t)((int64_t)(e) + 1023) << 52)) /* |e| < 1022 */
#endif

/* ---- division ---------------------------------------------------------------------------- */

/* a / b for call sites whose operands are finite, b is non-zero and a, b, a/b and 1/b lie well inside the normal range
 * (|exponent| < 500; a may also be +0).  On the host this IS the IEEE division.  On gfx950 the compiler expands `/` into
 * v_div_scale x2, v_rcp, two Newton steps on the reciprocal, q = a r, one residual correction, v_div_fmas, v_div_fixup;
 * for operands in that range the scaling steps multiply by 1 and the fix-up passes the quotient through, so the same
 * sequence without them returns the same bits with 8 instructions instead of 11.  (The march executes ~70 divisions per RK4 step.) */
#if defined(__HIP_DEVICE_COMPILE__)
DM_FN double dm_div(double a, double b) {
  double r = __builtin_amdgcn_rcp(b);
  double e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  double q = a * r;
  e = DM_FNMA_VVV(b, q, a);
  return __builtin_fma(e, r, q);
}
#else
DM_FN double dm_div(double a, double b) { return a / b; }
#endif

/* a / b from a SEED r0 of 1/b with |1 - b r0| <= 2^-20 — a bound the CALLER holds (a certificate, atm_certify): two Newton steps take
 * the seed to 2^-80 before rounding, the class of reciprocal dm_div's own two steps from v_rcp_f64 produce, and the quotient
 * correction is dm_div's.  7 full-rate instructions in place of the quarter-rate v_rcp_f64 + 7.  Host: the IEEE division.
 * Same operand range as dm_div. */
#if defined(__HIP_DEVICE_COMPILE__)
DM_FN double dm_div_seeded(double a, double b, double r0) {
  double e = __builtin_fma(-b, r0, 1.0);
  double r = __builtin_fma(r0, e, r0);
  e = __builtin_fma(-b, r, 1.0);
  r = __builtin_fma(r, e, r);
  double q = a * r;
  e = DM_FNMA_VVV(b, q, a);
  return __builtin_fma(e, r, q);
}
#else
DM_FN double dm_div_seeded(double a, double b, double r0) { (void)r0; return a / b; }
#endif

/* Three divisions a_i / b_i whose divisors are (expected to be) within 2^-20 of one another — the temperatures, and the
 * compressibilities, of the three evaluations n(h), n(h - eps), n(h + eps) of one ODE right-hand side.  On the host: three IEEE
 * divisions.  On gfx950: b_0's reciprocal is refined exactly as dm_div refines it and then SEEDS the two other reciprocals: with a
 * seed error |1 - b_i r_0| <= 2^-20 two Newton steps leave 2^-80 before rounding, the same class of reciprocal dm_div's own two
 * steps from v_rcp_f64 produce, and the quotient correction is dm_div's.  That saves the two quarter-rate v_rcp_f64 (the guard costs
 * what the two instructions save).  The guard is a wave vote; lanes that fail it (or hold NaN) send the whole wavefront through
 * dm_div.  Same operand range as dm_div.  tests/test_gpu_detmath.py checks the sequence against IEEE division like dm_div.
 * dm_div3_seeded: the same for call sites that HOLD the bound |1 - b_i / b_0| <= 2^-21 as a certificate (atm_certify's `tight`
 * segments) — no vote — and, optionally, a seed r0 for b_0's own reciprocal (have_seed: |1 - b_0 r0| <= 2^-20, likewise
 * certified) in place of its v_rcp_f64. */
#if defined(__HIP_DEVICE_COMPILE__)
DM_FN void dm_div3(double a0, double b0, double a1, double b1, double a2, double b2, double* q0, double* q1, double* q2) {
  double r = __builtin_amdgcn_rcp(b0);
  double e = __builtin_fma(-b0, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-b0, r, 1.0);
  r = __builtin_fma(r, e, r);
  double q = a0 * r;
  e = DM_FNMA_VVV(b0, q, a0);
  *q0 = __builtin_fma(e, r, q);
  double e1 = __builtin_fma(-b1, r, 1.0), e2 = __builtin_fma(-b2, r, 1.0);
  double m = __builtin_fmax(__builtin_fabs(e1), __builtin_fabs(e2));
  if (__all(m <= 9.5367431640625e-07)) { /* 2^-20; false for NaN */
    double r1 = __builtin_fma(r, e1, r);
    e1 = __builtin_fma(-b1, r1, 1.0);
    r1 = __builtin_fma(r1, e1, r1);
    q = a1 * r1;
    e1 = DM_FNMA_VVV(b1, q, a1);
    *q1 = __builtin_fma(e1, r1, q);
    double r2 = __builtin_fma(r, e2, r);
    e2 = __builtin_fma(-b2, r2, 1.0);
    r2 = __builtin_fma(r2, e2, r2);
    q = a2 * r2;
    e2 = DM_FNMA_VVV(b2, q, a2);
    *q2 = __builtin_fma(e2, r2, q);
  } else {
    *q1 = dm_div(a1, b1);
    *q2 = dm_div(a2, b2);
  }
}
DM_FN void dm_div3_seeded(double a0, double b0, double a1, double b1, double a2, double b2, int have_seed, double r0, double* q0, double* q1,
                          double* q2) {
  double r, e;
  if (have_seed) { /* compile-time at every call site */
    e = __builtin_fma(-b0, r0, 1.0);
    r = __builtin_fma(r0, e, r0);
  } else {
    r = __builtin_amdgcn_rcp(b0);
    e = __builtin_fma(-b0, r, 1.0);
    r = __builtin_fma(r, e, r);
  }
  e = __builtin_fma(-b0, r, 1.0);
  r = __builtin_fma(r, e, r);
  double q = a0 * r;
  e = DM_FNMA_VVV(b0, q, a0);
  *q0 = __builtin_fma(e, r, q);
  double e1 = __builtin_fma(-b1, r, 1.0), e2 = __builtin_fma(-b2, r, 1.0);
  double r1 = __builtin_fma(r, e1, r);
  e1 = __builtin_fma(-b1, r1, 1.0);
  r1 = __builtin_fma(r1, e1, r1);
  q = a1 * r1;
  e1 = DM_FNMA_VVV(b1, q, a1);
  *q1 = __builtin_fma(e1, r1, q);
  double r2 = __builtin_fma(r, e2, r);
  e2 = __builtin_fma(-b2, r2, 1.0);
  r2 = __builtin_fma(r2, e2, r2);
  q = a2 * r2;
  e2 = DM_FNMA_VVV(b2, q, a2);
  *q2 = __builtin_fma(e2, r2, q);
}
#else
DM_FN void dm_div3(double a0, double b0, double a1, double b1, double a2, double b2, double* q0, double* q1, double* q2) {
  *q0 = a0 / b0;
  *q1 = a1 / b1;
  *q2 = a2 / b2;
}
DM_FN void dm_div3_seeded(double a0, double b0, double a1, double b1, double a2, double b2, int have_seed, double r0, double* q0, double* q1,
                          double* q2) {
  (void)have_seed;
  (void)r0;
  *q0 = a0 / b0;
  *q1 = a1 / b1;
  *q2 = a2 / b2;
}
#endif

/* a / b given y = RN(1/b) (a correctly rounded reciprocal, e.g. tabulated on the host): two residual corrections.  After the
 * first, q1 is a faithful quotient; with y correctly rounded the second then yields RN(a/b) (Markstein's theorem), for operands
 * in the range described at dm_div.  5 instructions.  dm_div_r_seq is the sequence itself (tests/test_detmath.py checks it
 * against `/` on the host); dm_div_r is what call sites use: the sequence on the GPU, the plain division on the host. */
DM_FN double dm_div_r_seq(double a, double b, double y) {
  double q = a * y;
  double e = DM_FMA(-b, q, a);
  q = DM_FMA(e, y, q);
  e = DM_FMA(-b, q, a);
  return DM_FMA(e, y, q);
}
#if defined(__HIP_DEVICE_COMPILE__)
DM_FN double dm_div_r(double a, double b, double y) { return dm_div_r_seq(a, b, y); }
#else
DM_FN double dm_div_r(double a, double b, double y) { (void)y; return a / b; }
#endif

/* sqrt(x) for x in the normal range well away from its ends (2^-500 < x < 2^500; not 0).  The host takes the IEEE square root.
 * gfx950 expands sqrt into a scaling test, v_rsq, a coupled Newton iteration on (g ~ sqrt x, h ~ 1/(2 sqrt x)), two residual
 * corrections, an unscaling and a class fix-up for 0/inf; for x in that range the scaling is by 2^0 and the fix-up passes g
 * through, so the iteration alone returns the same bits with 10 instructions instead of 20. */
#if defined(__HIP_DEVICE_COMPILE__)
DM_FN double dm_sqrt_inrange(double x) {
  double y = __builtin_amdgcn_rsq(x);
  double g = x * y;
  double h = y * 0.5;
  double r = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, r, g);
  h = __builtin_fma(h, r, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
#else
DM_FN double dm_sqrt_inrange(double x) { return __builtin_sqrt(x); }
#endif

/* ---- sin / cos ------------------------------------------------------------------------- */

/* x = n*(pi/2) + (y0 + y1), |y0| <= ~pi/4.  Three-term Cody–Waite; the first two terms have 33
 * significant bits so n*term is exact for |n| < 2^20 (|x| < ~1.6e6, far beyond any angle here). */
DM_FN int dm_rem_pio2(double x, double* y0, double* y1) {
  const double invpio2 = 0.6366197723675814;
  const double p1 = 1.5707963267341256;     /* first 33 bits of pi/2 */
  const double p2 = 6.077100506303966e-11;  /* next 33 bits */
  const double p3 = 2.0222662487959506e-21; /* remainder, full double */
  double fn = dm_rint(x * invpio2);
  double r0 = x - fn * p1; /* exact */
  double w1 = fn * p2;     /* exact */
  /* TwoSum(r0, -w1) */
  double r1 = r0 - w1;
  double bb = r1 - r0;
  double e1 = (r0 - (r1 - bb)) + ((-w1) - bb);
  double tail = e1 - fn * p3;
  double a = r1 + tail;
  *y0 = a;
  *y1 = (r1 - a) + tail;
  /* |fn| < 2^31 for every reduced argument that is meaningful; saturate the rest (results are NaN or inaccurate anyway) */
  if (!(fn > -2147483000.0 && fn < 2147483000.0)) fn = 0.0;
  return (int)fn & 3;
}

DM_FN double dm_ksin(double x, double y) {
  const double S1 = -1.66666666666666324348e-01, S2 = 8.33333333332248946124e-03,
               S3 = -1.98412698298579493134e-04, S4 = 2.75573137070700676789e-06,
               S5 = -2.50507602534068634195e-08, S6 = 1.58969099521155010221e-10;
  double z = x * x;
  double v = z * x;
  double r = S2 + z * (S3 + z * (S4 + z * (S5 + z * S6)));
  return x - ((z * (0.5 * y - v * r) - y) - v * S1);
}

DM_FN double dm_kcos(double x, double y) {
  const double C1 = 4.16666666666666019037e-02, C2 = -1.38888888888741095749e-03,
               C3 = 2.48015872894767294178e-05, C4 = -2.75573143513906633035e-07,
               C5 = 2.08757232129817482790e-09, C6 = -1.13596475577881948265e-11;
  double z = x * x;
  double w = z * z;
  double r = z * (C1 + z * (C2 + z * C3)) + (w * w) * (C4 + z * (C5 + z * C6));
  double hz = 0.5 * z;
  double t = 1.0 - hz;
  return t + (((1.0 - t) - hz) + (z * r - x * y));
}

/* Straight-line: the reduction is exact for |x| <= pi/4 as well (n = 0, y0 = x, y1 = 0), so there is no
 * small-argument branch, and the quadrant is applied with selects.  inf/nan propagate as NaN. */
DM_FN void dm_sincos(double x, double* s, double* c) {
  double y0, y1, ks, kc, ss, cc;
  int n;
#if defined(__HIP_DEVICE_COMPILE__)
  /* every active lane within pi/4 (the arc angle of a marching step always is): the reduction below would return n = 0,
   * y0 = x, y1 = +0 exactly, so it is skipped — same bits, ~25 instructions fewer per call */
  if (__all(dm_fabs(x) <= 0.78539816339744828)) {
    *s = dm_ksin(x, 0.0);
    *c = dm_kcos(x, 0.0);
    return;
  }
#endif
  n = dm_rem_pio2(x, &y0, &y1);
  ks = dm_ksin(y0, y1);
  kc = dm_kcos(y0, y1);
  ss = (n & 1) ? kc : ks;
  cc = (n & 1) ? ks : kc;
  *s = (n & 2) ? -ss : ss;
  *c = ((n + 1) & 2) ? -cc : cc;
}
DM_FN double dm_sin(double x) {
  double s, c;
  dm_sincos(x, &s, &c);
  return s;
}
DM_FN double dm_cos(double x) {
  double s, c;
  dm_sincos(x, &s, &c);
  return c;
}
/* tan = sin/cos (<= 2 ulp); used only in per-frame set-up (rectilinear.rs:83, stepper start,
 * directional_calc.rs:110-112), never inside a marching loop. */
DM_FN double dm_tan(double x) {
  double s, c;
  dm_sincos(x, &s, &c);
  return s / c;
}

/* ---- atan / atan2 ---------------------------------------------------------------------- */

DM_FN double dm_atan(double x) {
  const double aT0 = 3.33333333333329318027e-01, aT1 = -1.99999999998764832476e-01,
               aT2 = 1.42857142725034663711e-01, aT3 = -1.11111104054623557880e-01,
               aT4 = 9.09088713343650656196e-02, aT5 = -7.69187620504482999495e-02,
               aT6 = 6.66107313738753120669e-02, aT7 = -5.83357013379057348645e-02,
               aT8 = 4.97687799461593236017e-02, aT9 = -3.65315727442169155270e-02,
               aT10 = 1.62858201153657823623e-02;
  double ax = dm_fabs(x);
  double hi = 0.0, lo = 0.0, t, z, w, s1, s2, res;
  int id;
  if (dm_isnan(x)) return x + x;
  if (ax >= 7.378697629483821e19) { /* 2^66 */
    res = 1.5707963267948966 + 6.123233995736766e-17;
    return dm_copysign(res, x);
  }
  if (ax < 0.4375) {
    if (ax < 7.450580596923828e-09) return x; /* 2^-27 */
    id = -1;
    t = x;
  } else if (ax < 1.1875) {
    if (ax < 0.6875) {
      id = 0; hi = 0.4636476090008061; lo = 2.2698777452961687e-17;
      t = dm_div(2.0 * ax - 1.0, 2.0 + ax);
    } else {
      id = 1; hi = 0.7853981633974483; lo = 3.061616997868383e-17;
      t = dm_div(ax - 1.0, ax + 1.0);
    }
  } else if (ax < 2.4375) {
    id = 2; hi = 0.982793723247329; lo = 1.3903311031230998e-17;
    t = dm_div(ax - 1.5, 1.0 + 1.5 * ax);
  } else {
    id = 3; hi = 1.5707963267948966; lo = 6.123233995736766e-17;
    t = dm_div(-1.0, ax); /* 2.4375 <= ax < 2^66 */
  }
  z = t * t;
  w = z * z;
  s1 = z * (aT0 + w * (aT2 + w * (aT4 + w * (aT6 + w * (aT8 + w * aT10)))));
  s2 = w * (aT1 + w * (aT3 + w * (aT5 + w * (aT7 + w * aT9))));
  if (id < 0) return t - t * (s1 + s2);
  res = hi - ((t * (s1 + s2) - lo) - t);
  return dm_copysign(res, x);
}

DM_FN double dm_atan2(double y, double x) {
  const double pi = 3.141592653589793, pi_lo = 1.2246467991473532e-16;
  const double pio2 = 1.5707963267948966, pio2_lo = 6.123233995736766e-17;
  const double pio4 = 0.7853981633974483;
  int ysign = (int)(dm_bits(y) >> 63);
  int xsign = (int)(dm_bits(x) >> 63);
  double ay = dm_fabs(y), ax = dm_fabs(x), z;
  int k;
  if (dm_isnan(x) || dm_isnan(y)) return x + y;
  if (ay == 0.0) { /* atan2(+-0, x) */
    if (!xsign) return y;
    return ysign ? -pi : pi;
  }
  if (ax == 0.0) return ysign ? -(pio2 + pio2_lo) : (pio2 + pio2_lo);
  if (dm_isinf(ax)) {
    if (dm_isinf(ay)) z = xsign ? 3.0 * pio4 : pio4;
    else z = xsign ? pi : 0.0;
    return ysign ? -z : z;
  }
  if (dm_isinf(ay)) return ysign ? -(pio2 + pio2_lo) : (pio2 + pio2_lo);
  /* the exponent difference decides the shortcuts, as in the classical algorithm */
  k = (int)((dm_hi(ay) >> 20) & 0x7ff) - (int)((dm_hi(ax) >> 20) & 0x7ff);
  if (k > 60) { /* |y/x| > 2^60 */
    z = pio2 + 0.5 * pio2_lo;
    return ysign ? -z : z;
  }
  if (xsign && k < -60) z = 0.0; /* 0 > |y|/x > -2^-60 */
  else z = dm_atan(ay / ax);
  if (!xsign) return ysign ? -z : z;
  z = pi - (z - pi_lo);
  return ysign ? -z : z;
}

/* ---- asin ------------------------------------------------------------------------------ */

DM_FN double dm_asin(double x) {
  const double pio2_hi = 1.5707963267948966, pio2_lo = 6.123233995736766e-17,
               pio4_hi = 0.7853981633974483;
  const double pS0 = 1.66666666666666657415e-01, pS1 = -3.25565818622400915405e-01,
               pS2 = 2.01212532134862925881e-01, pS3 = -4.00555345006794114027e-02,
               pS4 = 7.91534994289814532176e-04, pS5 = 3.47933107596021167570e-05,
               qS1 = -2.40339491173441421878e+00, qS2 = 2.02094576023350569471e+00,
               qS3 = -6.88283971605453293030e-01, qS4 = 7.70381505559019352791e-02;
  double ax = dm_fabs(x), t, p, q, w, s, c, r, res;
  if (dm_isnan(x)) return x + x;
  if (ax >= 1.0) {
    if (ax == 1.0) return x * pio2_hi + x * pio2_lo;
    return (x - x) / (x - x); /* NaN, as libm */
  }
  if (ax < 0.5) {
    if (ax < 7.450580596923828e-09) return x; /* 2^-27 */
    t = x * x;
    p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
    q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
    return x + x * dm_div(p, q);
  }
  w = 1.0 - ax;
  t = w * 0.5;
  p = t * (pS0 + t * (pS1 + t * (pS2 + t * (pS3 + t * (pS4 + t * pS5)))));
  q = 1.0 + t * (qS1 + t * (qS2 + t * (qS3 + t * qS4)));
  s = dm_sqrt_inrange(t); /* 2^-54 <= t <= 0.25 */
  if (ax >= 0.975) {
    w = dm_div(p, q);
    res = pio2_hi - (2.0 * (s + s * w) - pio2_lo);
  } else {
    w = dm_from_bits(dm_bits(s) & 0xffffffff00000000ULL);
    c = dm_div(t - w * w, s + w);
    r = dm_div(p, q);
    p = 2.0 * s * r - (pio2_lo - 2.0 * c);
    q = pio4_hi - 2.0 * w;
    res = pio4_hi - (p - q);
  }
  return dm_copysign(res, x);
}

/* Where the tables of exp and log are read from: the constant arrays of detmath_tables.h, or — in a GPU translation unit that
 * defines DM_TABLES_LDS as a __shared__ double[768] which each of its kernels fills first (log rows, then exp rows) — from LDS. */
/* u = ((hi + 0x1000) & ~0x1fff) - 0x3fe6a000: the constant's low 13 bits are zero, so it commutes with the mask — one addition
 * instead of two (integer identity modulo 2^32: the same u on every platform).  The row of u is table entry (u >> 13) & 127; the
 * LDS form addresses it by its byte offset (u >> 8) & 0xfe0 directly (rows are 32 bytes): two integer instructions, not three. */
#define DM_LOG_CENTRE(hi) ((int32_t)(((hi) + (0x1000u - 0x3fe6a000u)) & 0xffffe000u))
#define DM_LOG_Z_HI(hi, u, k) ((hi) - ((uint32_t)(u) & 0xfff00000u)) /* high word of z = x 2^-k */
#if defined(__HIP_DEVICE_COMPILE__) && defined(DM_TABLES_LDS)
#define DM_LOG_ROW(i) (&DM_TABLES_LDS[4 * (i)])
#define DM_LOG_ROW_OF(u) ((const double*)((const char*)DM_TABLES_LDS + (((uint32_t)(u) >> 8) & 0xfe0u)))
#define DM_EXP_ROW(j) (&DM_TABLES_LDS[512 + 2 * (j)])
#else
#define DM_LOG_ROW(i) DM_LOG_TAB[i]
#define DM_LOG_ROW_OF(u) DM_LOG_TAB[((u) >> 13) & 127]
#define DM_EXP_ROW(j) DM_EXP_TAB[j]
#endif

/* ---- exp / log / pow ------------------------------------------------------------------- */

DM_FN double dm_exp_slow(double x) { /* |x| > 700, nan (K.C. Ng's form; not on any hot path) */
  const double ln2hi = 6.93147180369123816490e-01, ln2lo = 1.90821492927058770002e-10,
               invln2 = 1.44269504088896338700e+00;
  const double P1 = 1.66666666666666019037e-01, P2 = -2.77777777770155933842e-03,
               P3 = 6.61375632143793436117e-05, P4 = -1.65339022054652515390e-06,
               P5 = 4.13813679705723846039e-08;
  double hi, lo, r, t, c, y, fk;
  int64_t k;
  if (dm_isnan(x)) return x + x;
  if (x > 709.782712893384) return dm_inf();
  if (x < -745.1332191019411) return 0.0;
  if (dm_fabs(x) < 3.725290298461914e-09) return 1.0 + x; /* 2^-28 */
  fk = dm_rint(x * invln2);
  hi = x - fk * ln2hi;
  lo = fk * ln2lo;
  r = hi - lo;
  k = (int64_t)fk;
  t = r * r;
  c = r - t * (P1 + t * (P2 + t * (P3 + t * (P4 + t * P5))));
  y = 1.0 - ((lo - (r * c) / (2.0 - c)) - hi);
  if (k >= -1021 && k <= 1023) return y * dm_from_bits((uint64_t)(k + 1023) << 52);
  if (k > 1023) return (y * dm_from_bits((uint64_t)(k - 1 + 1023) << 52)) * 2.0;
  /* gradual underflow: scale in two exact-power steps */
  return (y * dm_from_bits((uint64_t)(k + 1000 + 1023) << 52)) * dm_from_bits((uint64_t)(1023 - 1000) << 52);
}

/* exp(x) = 2^e 2^(j/128) e^r with x = (128 e + j) ln2/128 + r, |r| <= ln2/256 (Tang's table-driven reduction; the table
 * holds 2^(j/128) as a double-double, e^r - 1 is the degree-5 Taylor polynomial: r^6/720 < 2^-60).  Both reduction steps are
 * single FMAs (the first is exact: kd ln2N_hi and x share their bits above 2^-60).  One straight line for |x| <= 700
 * (for |x| <= ln2/256 kd = 0 and r = x exactly); everything else goes to dm_exp_slow.  < 0.6 ulp. */
DM_FN double dm_exp_main(double x) { /* |x| <= 700 */
  double kd, r, r2, q, p, s, y;
  const double* t;
  int32_t ki, e;
  kd = dm_rint(x * DM_INVLN2N);
  ki = (int32_t)kd;
  r = DM_FMA(-kd, DM_LN2N_HI, x);
  r = DM_FMA(-kd, DM_LN2N_LO, r);
  t = DM_EXP_ROW(ki & 127);
  e = ki >> 7;
  r2 = r * r;
  q = DM_FMA_VSV(r, 8.33333333333333333e-03, 4.16666666666666667e-02);
  q = DM_FMA_VVS(r, q, 1.66666666666666667e-01);
  q = DM_FMA(r, q, 0.5);
  p = DM_FMA(r2, q, r);
  s = DM_FMA(t[0], p, t[1]);
  y = t[0] + s;
  return DM_SCALBN(y, e); /* |e| <= 1010: the result is normal */
}
DM_FN int dm_exp_in_main_range(double x) { return dm_fabs(x) <= 700.0; } /* false for NaN */
DM_FN double dm_exp(double x) {
  if (!dm_exp_in_main_range(x)) return dm_exp_slow(x);
  return dm_exp_main(x);
}

/* log(x) for positive normal x: x = 2^k z with z within half a step of one of 128 centres c (7 mantissa bits, from 0.707 to
 * 1.406; the centre of the interval around 1 is exactly 1), r = z/c - 1 by one FMA with the tabulated 1/c (|r| <= 2^-8),
 * log x = k ln2 + log c + log1p(r); log c = -log(invc) is tabulated as a double-double, log1p(r) is the degree-7 Taylor
 * polynomial (r^8/8 < 2^-67).  k ln2_hi and logc_hi are multiples of 2^-42, so w below is exact; the rest of the sum is
 * accumulated as hi + lo like a double-double.  < 0.85 ulp (the worst cases sit next to the interval around 1). */
DM_FN double dm_log_core(double x, int32_t k0) {
  uint64_t ix = dm_bits(x);
  uint32_t hi = (uint32_t)(ix >> 32);
  int32_t u = DM_LOG_CENTRE(hi); /* nearest centre, relative to 0.70703125 */
  int32_t k = u >> 20;
  const double* t = DM_LOG_ROW_OF(u);
  double z = dm_from_bits((ix & 0xffffffffULL) | ((uint64_t)DM_LOG_Z_HI(hi, u, k) << 32));
  double r = DM_FMA(z, t[0], -1.0);
  double kd = (double)(k + k0);
  double w = DM_FMA(kd, DM_LN2_HI, t[1]);
  double h = w + r;
  double l = (w - h) + r;
  double r2 = r * r, p;
  l = l + DM_FMA(kd, DM_LN2_LO, t[2]);
  p = DM_FMA(r, 1.42857142857142857e-01, -1.66666666666666667e-01);
  p = DM_FMA(r, p, 0.2);
  p = DM_FMA(r, p, -0.25);
  p = DM_FMA(r, p, 3.33333333333333333e-01);
  p = DM_FMA(r, p, -0.5);
  return DM_FMA(r2, p, l) + h;
}
/* The same reduction and polynomial without the double-double accumulation, for pow(x, y) = exp(y log x): w = k ln2_hi + logc_hi is
 * exact, r + r^2 p(r) and the low words are tiny beside it (or, in the interval around 1 where w = 0, ARE the result), so the only
 * rounding of weight is the final addition: <= 1.1 ulp instead of 0.74, three instructions fewer — and pow's own error is dominated by
 * the rounding of y log x anyway.  Used by dm_pow (and by the stepping kernels' three-point form) only; dm_log keeps the accurate core. */
DM_FN double dm_log_core_pow(double x) {
  uint64_t ix = dm_bits(x);
  uint32_t hi = (uint32_t)(ix >> 32);
  int32_t u = DM_LOG_CENTRE(hi);
  int32_t k = u >> 20;
  const double* t = DM_LOG_ROW_OF(u);
  double z = dm_from_bits((ix & 0xffffffffULL) | ((uint64_t)DM_LOG_Z_HI(hi, u, k) << 32));
  double r = DM_FMA(z, t[0], -1.0);
  double kd = (double)k;
  double w = DM_FMA(kd, DM_LN2_HI, t[1]);
  double lo = DM_FMA(kd, DM_LN2_LO, t[2]);
  double r2 = r * r, p;
  p = DM_FMA_VSV(r, 1.42857142857142857e-01, -1.66666666666666667e-01);
  p = DM_FMA_VVS(r, p, 0.2);
  p = DM_FMA_VVS(r, p, -0.25);
  p = DM_FMA_VVS(r, p, 3.33333333333333333e-01);
  p = DM_FMA(r, p, -0.5);
  return w + (DM_FMA(r2, p, r) + lo);
}
/* Three-point forms for the stepping kernels: n(h), n(h - eps), n(h + eps) of one ODE right-hand side evaluate log and exp at
 * arguments that a certificate (atm_certify, TIGHT segments) places within 2^-21 of one another.  Then the three calls of
 * dm_log_core_pow almost always read the SAME table row with the SAME exponent, and the three calls of dm_exp_main the same 2^(j/128)
 * entry: each form computes that shared part once — u, k, the row address, two table reads, kd, k ln2 + log c (log: 8 VALU
 * instructions and 2 LDS reads per extra point; exp: 6 and 1) — and returns, wave-uniformly, whether it could:
 *   log: the high words of the arguments differ by at most one unit (2^-21 relative is half a unit), so the centres agree unless
 *        (hi_0 + 0x1000) & 0x1fff is 0 or 0x1fff — the centre argument sits on the edge of its table interval;
 *   exp: |e_i - e_0| 128 / ln2 <= margin (the segment's bound, `thr` = 1/2 - margin), so rint(e_i 128 / ln2) = rint(e_0 128 / ln2)
 *        unless the centre's own product lies within margin of a half-integer.
 * One lane at an edge sends its wavefront through the plain calls (the caller's other branch); the values are those of the plain
 * calls bit for bit either way — same operations on the same operands.  GPU only; tests/test_gpu_detmath.py (POW3_SHARED). */
#if defined(__HIP_DEVICE_COMPILE__)
DM_FN int dm_log3_core_pow_shared(double x0, double x1, double x2, double* l0, double* l1, double* l2) {
  const uint64_t i0 = dm_bits(x0), i1 = dm_bits(x1), i2 = dm_bits(x2);
  const uint32_t h0 = (uint32_t)(i0 >> 32), h1 = (uint32_t)(i1 >> 32), h2 = (uint32_t)(i2 >> 32);
  const uint32_t low = (h0 + 0x1000u) & 0x1fffu;
  if (!__all(low - 1u < 0x1ffeu)) return 0; /* 1 .. 0x1ffe */
  const int32_t u = DM_LOG_CENTRE(h0);
  const int32_t k = u >> 20;
  const double* t = DM_LOG_ROW_OF(u);
  const uint32_t base = (uint32_t)u & 0xfff00000u;
  const double z0 = dm_from_bits((i0 & 0xffffffffULL) | ((uint64_t)(h0 - base) << 32));
  const double z1 = dm_from_bits((i1 & 0xffffffffULL) | ((uint64_t)(h1 - base) << 32));
  const double z2 = dm_from_bits((i2 & 0xffffffffULL) | ((uint64_t)(h2 - base) << 32));
  const double invc = t[0];
  const double kd = (double)k;
  const double w = DM_FMA(kd, DM_LN2_HI, t[1]);
  const double lo = DM_FMA(kd, DM_LN2_LO, t[2]);
  const double r0 = DM_FMA(z0, invc, -1.0), r1 = DM_FMA(z1, invc, -1.0), r2 = DM_FMA(z2, invc, -1.0);
  double p0, p1, p2;
  p0 = DM_FMA_VSV(r0, 1.42857142857142857e-01, -1.66666666666666667e-01);
  p1 = DM_FMA_VSV(r1, 1.42857142857142857e-01, -1.66666666666666667e-01);
  p2 = DM_FMA_VSV(r2, 1.42857142857142857e-01, -1.66666666666666667e-01);
  p0 = DM_FMA_VVS(r0, p0, 0.2), p1 = DM_FMA_VVS(r1, p1, 0.2), p2 = DM_FMA_VVS(r2, p2, 0.2);
  p0 = DM_FMA_VVS(r0, p0, -0.25), p1 = DM_FMA_VVS(r1, p1, -0.25), p2 = DM_FMA_VVS(r2, p2, -0.25);
  p0 = DM_FMA_VVS(r0, p0, 3.33333333333333333e-01), p1 = DM_FMA_VVS(r1, p1, 3.33333333333333333e-01), p2 = DM_FMA_VVS(r2, p2, 3.33333333333333333e-01);
  p0 = DM_FMA(r0, p0, -0.5), p1 = DM_FMA(r1, p1, -0.5), p2 = DM_FMA(r2, p2, -0.5);
  *l0 = w + (DM_FMA(r0 * r0, p0, r0) + lo);
  *l1 = w + (DM_FMA(r1 * r1, p1, r1) + lo);
  *l2 = w + (DM_FMA(r2 * r2, p2, r2) + lo);
  return 1;
}
DM_FN int dm_exp3_main_shared(double e0, double e1, double e2, double thr, double* y0, double* y1, double* y2) {
  const double t0 = e0 * DM_INVLN2N;
  const double kd = dm_rint(t0);
  if (!__all(dm_fabs(t0 - kd) <= thr)) return 0; /* false for NaN */
  const int32_t ki = (int32_t)kd;
  const double* t = DM_EXP_ROW(ki & 127);
  const int32_t e = ki >> 7;
  const double th = t[0], tl = t[1];
  double r0 = DM_FMA(-kd, DM_LN2N_HI, e0), r1 = DM_FMA(-kd, DM_LN2N_HI, e1), r2 = DM_FMA(-kd, DM_LN2N_HI, e2);
  r0 = DM_FMA(-kd, DM_LN2N_LO, r0), r1 = DM_FMA(-kd, DM_LN2N_LO, r1), r2 = DM_FMA(-kd, DM_LN2N_LO, r2);
  double q0, q1, q2;
  q0 = DM_FMA_VSV(r0, 8.33333333333333333e-03, 4.16666666666666667e-02);
  q1 = DM_FMA_VSV(r1, 8.33333333333333333e-03, 4.16666666666666667e-02);
  q2 = DM_FMA_VSV(r2, 8.33333333333333333e-03, 4.16666666666666667e-02);
  q0 = DM_FMA_VVS(r0, q0, 1.66666666666666667e-01), q1 = DM_FMA_VVS(r1, q1, 1.66666666666666667e-01), q2 = DM_FMA_VVS(r2, q2, 1.66666666666666667e-01);
  q0 = DM_FMA(r0, q0, 0.5), q1 = DM_FMA(r1, q1, 0.5), q2 = DM_FMA(r2, q2, 0.5);
  const double p0 = DM_FMA(r0 * r0, q0, r0), p1 = DM_FMA(r1 * r1, q1, r1), p2 = DM_FMA(r2 * r2, q2, r2);
  *y0 = DM_SCALBN(th + DM_FMA(th, p0, tl), e);
  *y1 = DM_SCALBN(th + DM_FMA(th, p1, tl), e);
  *y2 = DM_SCALBN(th + DM_FMA(th, p2, tl), e);
  return 1;
}
#endif

DM_FN double dm_log_slow(double x) { /* nan, negative, zero, inf, subnormal */
  if (dm_isnan(x)) return x + x;
  if (x < 0.0) return (x - x) / (x - x);
  if (x == 0.0) return -dm_inf();
  if (dm_isinf(x)) return x;
  return dm_log_core(x * 18014398509481984.0, -54); /* subnormal: scale by 2^54 */
}
#if defined(__HIP_DEVICE_COMPILE__)
DM_FN int dm_log_in_main_range(double x) { return __builtin_amdgcn_class(x, 0x100); } /* positive normal: one v_cmp_class */
#else
DM_FN int dm_log_in_main_range(double x) { return x >= 2.2250738585072014e-308 && x <= 1.7976931348623157e308; }
#endif
DM_FN double dm_log(double x) {
  if (!dm_log_in_main_range(x)) return dm_log_slow(x);
  return dm_log_core(x, 0);
}

/* x > 0 only (barometric formula: base = T/T_b in (0, 2)).  <= 3 ulp for |y ln x| <= 4. */
DM_FN double dm_pow(double x, double y) { return dm_exp(y * (dm_log_in_main_range(x) ? dm_log_core_pow(x) : dm_log_slow(x))); }

#endif /* ATMRT_DETMATH_H */

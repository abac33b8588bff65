// atmrt_core.h — f64 numerics of the ray-marching path, shared by the HIP kernels and by the
// library's host-side set-up code (atmosphere table, distance table).  Everything is
// __host__ __device__ and built on detmath.h so that host set-up and device kernels agree to the
// bit.  Reference citations are relative to /root/reference.
//
// Compile with -ffp-contract=off: the CPU checker (oracle/) executes the same operation sequence
// without FMA contraction and results are compared bit-for-bit.
#pragma once
#include <cstddef>

#include <stdint.h>

#include <vector>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define ATMRT_HD __host__ __device__ inline __attribute__((always_inline))
#define DM_FN __host__ __device__ static inline __attribute__((always_inline))
#else
#define ATMRT_HD inline
#endif
#include "detmath.h"

#include "../../include/atmrt.h"

namespace atmrt {

struct Vec3 {
  double x, y, z;
};
ATMRT_HD Vec3 v3(double x, double y, double z) { return Vec3{x, y, z}; }
ATMRT_HD Vec3 operator+(Vec3 a, Vec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
ATMRT_HD Vec3 operator-(Vec3 a, Vec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
ATMRT_HD Vec3 operator-(Vec3 a) { return v3(-a.x, -a.y, -a.z); }
ATMRT_HD Vec3 operator*(Vec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
ATMRT_HD Vec3 operator/(Vec3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
ATMRT_HD double dot(Vec3 a, Vec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
ATMRT_HD Vec3 cross(Vec3 a, Vec3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}

// ---------------------------------------------------------------------------------------------
// Atmosphere + refractive index (crate atm-refraction 0.6, source absent: published models —
// US Standard Atmosphere 1976 layers, hydrostatic ideal gas, Ciddor 1996 dry air).
// EVALUATION ORDER: oracle/atmosphere.c and oracle/stepper.c fix it (the crate's is unknown) and every function here follows them
// operation for operation — on a Linear segment x = T / tb = fma(lapse / tb, h - hb, 1), T = tb x; Z by two nested fused
// multiply-adds; dn = (n(h + eps) - n(h - eps)) * 50; the stepper's stage points and sums fused, its spherical right-hand side over
// one denominator.
// ---------------------------------------------------------------------------------------------

// The atmosphere compiled into SEGMENTS: a Linear function is one segment, a Spline contributes one segment per knot
// interval (plus linear continuations outside its knots).  In a segment T(h) = tb + c1 dh + c2 dh^2 + c3 dh^3 with
// dh = h - hb.  Linear segments (cubic == 0) use the closed-form hydrostatic pressure, cubic ones a 5-point
// Gauss-Legendre quadrature of dh/T.
// Any number of segments (the reference's AtmosphereDef holds `Vec`s: README.md:283-323, params.rs:453-454): the table is a
// 32-byte header followed, in the same allocation, by its n segments — one 160-byte record each, so the parameters of the
// wave-uniform hinted layer are one run of scalar loads and a per-lane layer is one gather base.
struct AtmSeg {
  // ---- the first 80 bytes are what one RK4 stage of the marching kernels reads on its normal path: one run of scalar loads ----
  // the TIGHT part [tight_lo, tight_hi) of the certified interval (empty: lo = +inf): the kernels vote on it first — the votes of
  // dm_div3 and the v_rcp_f64 of Z and n go — and on [safe_lo, safe_hi) only when a lane is outside it
  double tight_lo, tight_hi;
  double hb;    // reference altitude of the segment
  double tb;    // temperature at hb
  double gtb;   // lapse / tb: x = T / tb = fma(gtb, h - hb, 1)
  double ptb;   // pb / tb: the density term p / T at hb
  double expo1; // the exponent of p / T = ptb x^expo1: expo - 1; on an isothermal segment p / T = ptb exp(expo1 (h - hb)), expo1 = expo
  // TIGHT segments: 1/2 - margin, where margin bounds |e_i - e_0| 128 / ln2 for the arguments e of exp at the three evaluation
  // points of one right-hand side — dm_exp3_main_shared shares its table row among them while the centre's product is at most
  // this far from an integer (negative: never)
  double exp_thr;
  double k_refr; // a copy of AtmTable::k_refr (atm_certify): the stage needs no second load from the header
  // ATM_SEG_ISOTHERMAL: lapse == 0 (a scalar integer test in the kernels, where the double compare costs a VALU slot per stage).
  // ATM_SEG_TIGHT (atm_certify): over [tight_lo, tight_hi) the three evaluation points of one ODE right-hand side (1 cm apart) have
  // log arguments and compressibilities within 2^-21 of one another, and n - 1 and |1 - Z| stay below 2^-10.5 — so dm_div3 needs no
  // vote on its seeds, 2 - Z seeds the reciprocal of Z and 1 - (n - 1) that of n (dm_div3_seeded, dm_div_seeded).
  int32_t flags;
  int32_t cubic; // 1: a knot interval of a Spline temperature function
  // ---- the rest: the fall-back vote, the layer search, Spline segments, the host ----
  // atm_certify: the part [safe_lo, safe_hi) of the segment over which T, p, p/T, Z and n are PROVEN to stay inside the operand range of
  // the GPU's division / square-root shortcuts (detmath.h); empty (lo = +inf) when nothing can be proven.  An evaluation outside it
  // takes the IEEE operations, so a pathological atmosphere (a spline that overshoots to 30 K, a pressure of 1e308 Pa) is slower but
  // still bit-identical to the host.
  double safe_lo, safe_hi;
  double from;  // segment k >= 1 applies for h >= from
  double pb;    // pressure at hb
  double lapse; // c1 = dT/dh at hb
  double expo;  // linear: lapse != 0 ? -g0 M/(R lapse) : -g0 M/(R tb);  cubic: -g0 M/R
  double c2;
  double c3;
  double _pad[2];
};
constexpr int32_t ATM_SEG_ISOTHERMAL = 1, ATM_SEG_TIGHT = 2;
struct AtmTable {
  int32_t n;
  int32_t _pad;
  double k_refr;         // (n - 1) = k_refr * (p/T) / Z
  double alt_lo, alt_hi; // the altitudes at which a path-length step may use the shortcuts (atm_certify)
  ATMRT_HD const AtmSeg& seg(int k) const { return reinterpret_cast<const AtmSeg*>(this + 1)[k]; }
  ATMRT_HD AtmSeg& seg(int k) { return reinterpret_cast<AtmSeg*>(this + 1)[k]; }
};
static_assert(sizeof(AtmTable) == 32 && sizeof(AtmSeg) == 160 && offsetof(AtmSeg, k_refr) == 64, "device and host read the table as header + records");
// the table in the constant address space (it is read-only for a whole launch): wave-uniform indices become scalar loads
#if defined(__HIPCC__)
typedef const __attribute__((address_space(4))) AtmTable* AtmConstTable;
typedef const __attribute__((address_space(4))) AtmSeg* AtmConstSeg;
__device__ __forceinline__ AtmConstTable atm_const_table(const AtmTable& a) { return (AtmConstTable)(uintptr_t)&a; }
__device__ __forceinline__ AtmConstSeg atm_const_seg(const AtmTable& a, int k) { return (AtmConstSeg)((uintptr_t)&a + sizeof(AtmTable)) + k; }
#endif

// Atmosphere::layer: the last segment k >= 1 with h >= from, else 0 (from is non-decreasing over k >= 1; a NaN altitude falls to 0).
// Short tables are searched from the top like the layer list of a physical atmosphere; long ones (a radiosonde spline) by bisection —
// the same index either way.
ATMRT_HD int atm_layer(const AtmTable& a, double h) {
  const int n = a.n;
  if (n <= 12) {
    for (int k = n - 1; k >= 1; k--)
      if (h >= a.seg(k).from) return k;
    return 0;
  }
  int lo = 0, hi = n; // invariant: (lo == 0 or h >= from[lo]) and (hi == n or !(h >= from[hi]))
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (h >= a.seg(mid).from) lo = mid;
    else hi = mid;
  }
  return lo;
}

ATMRT_HD double seg_temperature(double tb, double c1, double c2, double c3, double dh) {
  return tb + dh * (c1 + dh * (c2 + dh * c3));
}

// integral of dh'/T(h') from hb to hb + dh over a cubic segment: 5-point Gauss-Legendre on [0, dh]
ATMRT_HD double seg_inv_t_integral(double tb, double c1, double c2, double c3, double dh) {
  const double x1 = 0.5384693101056831, x2 = 0.9061798459386640;
  const double w0 = 0.5688888888888889, w1 = 0.4786286704993665, w2 = 0.2369268850561891;
  double half = 0.5 * dh;
  double s = w0 / seg_temperature(tb, c1, c2, c3, half);
  s += w1 / seg_temperature(tb, c1, c2, c3, half - half * x1);
  s += w1 / seg_temperature(tb, c1, c2, c3, half + half * x1);
  s += w2 / seg_temperature(tb, c1, c2, c3, half - half * x2);
  s += w2 / seg_temperature(tb, c1, c2, c3, half + half * x2);
  return half * s;
}

// p(h) / pb of segment k
ATMRT_HD double seg_pressure_ratio(int cubic, double hb, double tb, double gtb, double c1, double c2, double c3, double expo, double h) {
  if (cubic) return dm_exp(expo * seg_inv_t_integral(tb, c1, c2, c3, h - hb));
  if (c1 != 0.0) return dm_pow(DM_FMA(gtb, h - hb, 1.0), expo);
  return dm_exp(expo * (h - hb));
}
ATMRT_HD double atm_pressure_ratio(const AtmTable& a, int k, double h) {
  return seg_pressure_ratio(a.seg(k).cubic, a.seg(k).hb, a.seg(k).tb, a.seg(k).gtb, a.seg(k).lapse, a.seg(k).c2, a.seg(k).c3, a.seg(k).expo, h);
}
ATMRT_HD double atm_seg_temperature(const AtmTable& a, int k, double h) {
  if (a.seg(k).cubic) return seg_temperature(a.seg(k).tb, a.seg(k).lapse, a.seg(k).c2, a.seg(k).c3, h - a.seg(k).hb);
  return a.seg(k).tb * DM_FMA(a.seg(k).gtb, h - a.seg(k).hb, 1.0);
}

// Host storage of a table: the header and its records in one block (what prepare_frame uploads as it is).
struct AtmTableBuf {
  std::vector<uint64_t> raw;
  AtmTable& table() { return *reinterpret_cast<AtmTable*>(raw.data()); }
  const AtmTable& table() const { return *reinterpret_cast<const AtmTable*>(raw.data()); }
  size_t bytes() const { return sizeof(AtmTable) + (size_t)table().n * sizeof(AtmSeg); }
  void alloc(size_t segments) { raw.assign((sizeof(AtmTable) + (segments ? segments : 1) * sizeof(AtmSeg)) / sizeof(uint64_t), 0); }
};

// Atmosphere::from_def (params.rs:514).  Returns 0, or a negative code: -1 bad counts or kinds, -2 altitudes not increasing,
// -3 no temperature anchor (all Linear without a fixed point), -4 bad spline.  Any number of functions and of spline points.
inline int atm_compile(const atmrt_atmosphere_t& def, double wavelength, AtmTableBuf& buf) {
  const double gmr = 9.80665 * 0.0289644 / 8.31432;
  const int nf = def.n_functions;
  if (nf < 1 || !def.functions) return -1;
  size_t cap = 0; // a Linear function is one segment, a Spline at most its knot intervals + the two linear continuations
  for (int j = 0; j < nf; j++) {
    const atmrt_temp_function_t& fn = def.functions[j];
    if (fn.kind == ATMRT_TEMP_LINEAR) cap += 1;
    else if (fn.kind == ATMRT_TEMP_SPLINE) {
      if (fn.n_points < 2 || !fn.point_altitude || !fn.point_temperature) return -4;
      cap += (size_t)fn.n_points + 1;
    } else return -1;
  }
  if (cap > (size_t)1 << 24) return -1;
  buf.alloc(cap);
  AtmTable& out = buf.table();
  for (int j = 2; j < nf; j++)
    if (!(def.functions[j].altitude > def.functions[j - 1].altitude)) return -2;
  // ---- segments, function by function; `owner` remembers which function a segment belongs to
  int n = 0;
  std::vector<int> owner(cap), first_seg((size_t)nf + 1);
  std::vector<char> anchored((size_t)nf);
  for (int j = 0; j < nf; j++) {
    const atmrt_temp_function_t& fn = def.functions[j];
    const bool has_lo = j > 0, has_hi = j + 1 < nf;
    const double lo = has_lo ? fn.altitude : 0.0, hi = has_hi ? def.functions[j + 1].altitude : 0.0;
    first_seg[j] = n;
    anchored[j] = false;
    if (fn.kind == ATMRT_TEMP_LINEAR) {
      owner[n] = j;
      out.seg(n).from = lo;
      out.seg(n).lapse = fn.gradient;
      n++;
      continue;
    }
    if (fn.kind != ATMRT_TEMP_SPLINE) return -1;
    const int np = fn.n_points;
    if (np < 2) return -4;
    const double* x = fn.point_altitude;
    const double* y = fn.point_temperature;
    for (int i = 1; i < np; i++)
      if (!(x[i] > x[i - 1])) return -4;
    // second derivatives m[i] of the interpolating cubic spline (Thomas algorithm)
    std::vector<double> m((size_t)np), cp((size_t)np), dp((size_t)np);
    {
      double b0, c0, d0, an, bn, dn;
      if (fn.boundary == ATMRT_SPLINE_DERIVATIVES) {
        double h0 = x[1] - x[0], hn = x[np - 1] - x[np - 2];
        b0 = 2.0 * h0; c0 = h0; d0 = 6.0 * ((y[1] - y[0]) / h0 - fn.bc[0]);
        an = hn; bn = 2.0 * hn; dn = 6.0 * (fn.bc[1] - (y[np - 1] - y[np - 2]) / hn);
      } else {
        b0 = 1.0; c0 = 0.0; d0 = fn.boundary == ATMRT_SPLINE_SECOND_DERIVATIVES ? fn.bc[0] : 0.0;
        an = 0.0; bn = 1.0; dn = fn.boundary == ATMRT_SPLINE_SECOND_DERIVATIVES ? fn.bc[1] : 0.0;
      }
      cp[0] = c0 / b0;
      dp[0] = d0 / b0;
      for (int i = 1; i < np; i++) {
        double ai, bi, ci, di;
        if (i < np - 1) {
          double hl = x[i] - x[i - 1], hr = x[i + 1] - x[i];
          ai = hl; bi = 2.0 * (hl + hr); ci = hr;
          di = 6.0 * ((y[i + 1] - y[i]) / hr - (y[i] - y[i - 1]) / hl);
        } else {
          ai = an; bi = bn; ci = 0.0; di = dn;
        }
        double den = bi - ai * cp[i - 1];
        cp[i] = ci / den;
        dp[i] = (di - ai * dp[i - 1]) / den;
      }
      m[np - 1] = dp[np - 1];
      for (int i = np - 2; i >= 0; i--) m[i] = dp[i] - cp[i] * m[i + 1];
    }
    // linear continuation below the first knot
    if (!has_lo || lo < x[0]) {
      double hh = x[1] - x[0];
      owner[n] = j;
      out.seg(n).from = lo;
      out.seg(n).hb = x[0];
      out.seg(n).tb = y[0];
      out.seg(n).lapse = (y[1] - y[0]) / hh - hh * (2.0 * m[0] + m[1]) / 6.0; // S'(x0)
      out.seg(n).gtb = out.seg(n).lapse / out.seg(n).tb;
      n++;
    }
    for (int i = 0; i + 1 < np; i++) {
      if (has_hi && x[i] >= hi) break;          // interval entirely above this function's range
      if (has_lo && x[i + 1] <= lo) continue;   // interval entirely below it
      double hh = x[i + 1] - x[i];
      owner[n] = j;
      out.seg(n).from = (has_lo && lo > x[i]) ? lo : x[i];
      out.seg(n).hb = x[i];
      out.seg(n).tb = y[i];
      out.seg(n).lapse = (y[i + 1] - y[i]) / hh - hh * (2.0 * m[i] + m[i + 1]) / 6.0;
      out.seg(n).c2 = m[i] / 2.0;
      out.seg(n).c3 = (m[i + 1] - m[i]) / (6.0 * hh);
      out.seg(n).cubic = 1;
      n++;
    }
    // linear continuation above the last knot
    if (!has_hi || hi > x[np - 1]) {
      double hh = x[np - 1] - x[np - 2];
      owner[n] = j;
      out.seg(n).from = (has_lo && lo > x[np - 1]) ? lo : x[np - 1]; // every knot below the function's own start: it still begins at `lo`
      out.seg(n).hb = x[np - 1];
      out.seg(n).tb = y[np - 1];
      out.seg(n).lapse = (y[np - 1] - y[np - 2]) / hh + hh * (m[np - 2] + 2.0 * m[np - 1]) / 6.0; // S'(x_last)
      out.seg(n).gtb = out.seg(n).lapse / out.seg(n).tb;
      n++;
    }
    anchored[j] = true;
  }
  first_seg[nf] = n;
  out.n = n;
  out.seg(0).from = 0.0; // segment 0 extends to -inf
  // ---- absolute temperature of the Linear functions: the fixed point, else continuity with an anchored neighbour
  if (def.has_temperature_fixed_point) {
    int jt = 0;
    for (int j = nf - 1; j >= 1; j--)
      if (def.temperature_altitude >= def.functions[j].altitude) { jt = j; break; }
    if (def.functions[jt].kind == ATMRT_TEMP_LINEAR && !anchored[jt]) {
      int k = first_seg[jt];
      out.seg(k).hb = jt == 0 ? def.temperature_altitude : out.seg(k).from;
      out.seg(k).tb = def.temperature - out.seg(k).lapse * (def.temperature_altitude - out.seg(k).hb);
      out.seg(k).gtb = out.seg(k).lapse / out.seg(k).tb;
      anchored[jt] = true;
    }
  }
  bool any = false;
  for (int j = 0; j < nf; j++) any = any || anchored[j];
  if (!any) return -3;
  for (int pass = 0; pass < nf; pass++) {
    for (int j = 0; j < nf; j++) {
      if (anchored[j]) continue;
      int k = first_seg[j];
      if (j > 0 && anchored[j - 1]) { // continuous with the function below at this function's start altitude
        int kl = first_seg[j] - 1;
        out.seg(k).hb = out.seg(k).from;
        out.seg(k).tb = atm_seg_temperature(out, kl, out.seg(k).from);
        out.seg(k).gtb = out.seg(k).lapse / out.seg(k).tb;
        anchored[j] = true;
      } else if (j + 1 < nf && anchored[j + 1]) { // continuous with the function above at its start altitude
        int ku = first_seg[j + 1];
        double top = def.functions[j + 1].altitude;
        out.seg(k).hb = j == 0 ? top : out.seg(k).from;
        out.seg(k).tb = atm_seg_temperature(out, ku, top) - out.seg(k).lapse * (top - out.seg(k).hb);
        out.seg(k).gtb = out.seg(k).lapse / out.seg(k).tb;
        anchored[j] = true;
      }
    }
  }
  for (int k = 0; k < n; k++) {
    out.seg(k).expo = out.seg(k).cubic ? -gmr : (out.seg(k).lapse != 0.0 ? -gmr / out.seg(k).lapse : -gmr / out.seg(k).tb);
    out.seg(k).gtb = out.seg(k).lapse / out.seg(k).tb; // (set where tb was: the anchoring above evaluates temperatures)
  }
  // ---- pressure: chain outwards from the pressure fixed point
  int jp = atm_layer(out, def.pressure_altitude);
  out.seg(jp).pb = def.pressure / atm_pressure_ratio(out, jp, def.pressure_altitude);
  for (int k = jp + 1; k < n; k++) {
    double pk = out.seg(k - 1).pb * atm_pressure_ratio(out, k - 1, out.seg(k).from); // p at the boundary, from below
    out.seg(k).pb = pk / atm_pressure_ratio(out, k, out.seg(k).from);
  }
  for (int k = jp - 1; k >= 0; k--) {
    double pk = out.seg(k + 1).pb * atm_pressure_ratio(out, k + 1, out.seg(k + 1).from); // p at the boundary, from above
    out.seg(k).pb = pk / atm_pressure_ratio(out, k, out.seg(k + 1).from);
  }
  for (int k = 0; k < n; k++) { // the density form of n(h) on Linear segments (oracle/atmosphere.c oracle_n)
    out.seg(k).ptb = out.seg(k).pb / out.seg(k).tb;
    out.seg(k).expo1 = out.seg(k).lapse != 0.0 ? out.seg(k).expo - 1.0 : out.seg(k).expo;
  }
  {
    const double k0 = 238.0185, k1 = 5792105.0, k2 = 57.362, k3 = 167917.0;
    const double xco2 = 450.0, pr1 = 101325.0, tr1 = 288.15, za = 0.9995922115, r = 8.314472;
    double lam_um = wavelength * 1.0e6;
    double s = 1.0 / (lam_um * lam_um);
    double ras = 1.0e-8 * (k1 / (k0 - s) + k3 / (k2 - s));
    double raxs = ras * (1.0 + 5.34e-7 * (xco2 - 450.0));
    double ma = 0.0289635 + 1.2011e-8 * (xco2 - 400.0);
    double rho_axs = pr1 * ma / (za * r * tr1);
    out.k_refr = raxs / rho_axs * ma / r;
  }
  return 0;
}

// ---- the certificate behind the GPU's shortcut divisions (AtmTable::safe_lo / safe_hi / alt_lo / alt_hi) ----------------------
// Can every evaluation of n(h), h in [lo, hi] inside segment k, keep its operands where dm_div / dm_div3 / dm_div_r return the
// IEEE quotient?  Sufficient conditions, with margins of hundreds of binades to the real limits (|exponent| < 500):
//   1 K <= T <= 1e5 K on the interval (and at the quadrature nodes of a cubic segment, which lie between hb and h);
//   1e-250 Pa <= p <= 1e250 Pa  (p is monotone in h while T > 0: the end points bound it; a cubic segment: p <= pb, and the
//   quadrature of 1/T is at most (h - hb) / Tmin);
//   |Z - 1| <= (p/T) |a0 + a1 t + a2 t^2| + (p/T)^2 d <= 1/2, bounded from max p / min T and max |t|;
//   n - 1 = k_refr (p/T) / Z <= 2^8.
// The predicate is monotone: a sub-interval of an interval that passes passes.
inline bool atm_interval_certified(const AtmTable& t, int k, double lo, double hi) {
  const double a0 = 1.58123e-6, a1 = -2.9331e-8, a2 = 1.1043e-10, d = 1.83e-11;
  const double tb = t.seg(k).tb, pb = t.seg(k).pb;
  if (!(tb >= 1.0 && tb <= 1.0e5) || !(pb >= 1.0e-250 && pb <= 1.0e250)) return false;
  double tmin, tmax, pmin, pmax;
  if (t.seg(k).cubic) {
    const double c1 = t.seg(k).lapse, c2 = t.seg(k).c2, c3 = t.seg(k).c3;
    const double d1 = hi - t.seg(k).hb; // T over [hb, hi]: the evaluation points and the quadrature nodes
    if (!(d1 >= 0.0) || !(lo >= t.seg(k).hb)) return false;
    tmin = tmax = tb;
    double cand[3] = {d1, -1.0, -1.0};
    if (c3 != 0.0) {
      const double disc = c2 * c2 - 3.0 * c3 * c1;
      if (disc >= 0.0) {
        const double r = dm_sqrt(disc);
        cand[1] = (-c2 + r) / (3.0 * c3);
        cand[2] = (-c2 - r) / (3.0 * c3);
      }
    } else if (c2 != 0.0) {
      cand[1] = -c1 / (2.0 * c2);
    }
    for (int i = 0; i < 3; i++) {
      if (!(cand[i] >= 0.0 && cand[i] <= d1)) continue;
      const double v = seg_temperature(tb, c1, c2, c3, cand[i]);
      tmin = v < tmin ? v : tmin;
      tmax = v > tmax ? v : tmax;
    }
    // the stationary points are located in floating point: allow for it
    tmin -= 1.0e-6 * (dm_fabs(tmin) + dm_fabs(tmax)) + 1.0e-6;
    tmax += 1.0e-6 * (dm_fabs(tmin) + dm_fabs(tmax)) + 1.0e-6;
    if (!(tmin >= 1.0 && tmax <= 1.0e5)) return false;
    pmax = pb;
    // p / pb = exp(e), e = expo * (integral of dh / T) in [expo d1 / tmin, 0]: at least -640, so that the certified evaluation may
    // call the main branch of exp unasked here too (refr_n_layer_inrange)
    if (!(t.seg(k).expo * d1 / tmin >= -640.0)) return false;
    pmin = pb * dm_exp(t.seg(k).expo * d1 / tmin); // expo < 0
  } else {
    if (t.seg(k).lapse != 0.0 && !(dm_fabs(t.seg(k).expo) <= 1.0e6)) return false; // a lapse rate below 4e-8 K/m: the exponent of pow runs away
    const double t0 = atm_seg_temperature(t, k, lo), t1 = atm_seg_temperature(t, k, hi);
    tmin = t0 < t1 ? t0 : t1;
    tmax = t0 < t1 ? t1 : t0;
    if (!(tmin >= 1.0 && tmax <= 1.0e5)) return false;
    // p / pb = exp(e) with e = expo * log(T / tb) (or expo * (h - hb) on an isothermal segment): monotone in h, so the end points
    // bound it.  Within 1e-280 .. 1e280 means |e| <= 645 at every point of the interval, and the exponent the kernels form, that of
    // p / T = ptb x^(expo - 1), differs from e by log x, |log x| <= 11.6: at most 657 < 700 — the certified evaluation may call
    // the main branch of exp without asking (refr_n_layer3) — and x = fma(gtb, h - hb, 1), monotone in h and with T = tb x in 1 .. 1e5 K
    // at both ends, is a positive normal number (>= 1e-5 (1 - 2^-52)): the main branch of log likewise.
    const double r0 = atm_pressure_ratio(t, k, lo), r1 = atm_pressure_ratio(t, k, hi);
    if (!(r0 >= 1.0e-280 && r0 <= 1.0e280 && r1 >= 1.0e-280 && r1 <= 1.0e280)) return false;
    const double p0 = pb * r0, p1 = pb * r1;
    pmin = p0 < p1 ? p0 : p1;
    pmax = p0 < p1 ? p1 : p0;
  }
  if (!(pmin >= 1.0e-250 && pmax <= 1.0e250)) return false;
  const double ptmax = pmax / tmin * 1.000001;
  const double tm0 = dm_fabs(tmin - 273.15), tm1 = dm_fabs(tmax - 273.15), tm = tm0 > tm1 ? tm0 : tm1;
  const double amax = a0 + tm * (dm_fabs(a1) + tm * a2);
  if (!(ptmax * amax + ptmax * ptmax * d <= 0.5)) return false;
  return t.k_refr * ptmax * 2.0 <= 256.0;
}

// The extra bounds of a TIGHT segment over its certified interval [lo, hi] (Linear segments only; atm_interval_certified holds):
//   (1) |lapse| eps / Tmin <= 2^-21: the arguments x = fma(gtb, h - hb, 1) of log at h, h -+ eps differ by |gtb| eps = |lapse| eps / tb,
//       relative to x = T / tb that is |lapse| eps / T; with the rounding of each x (2^-53 absolute, x >= 1e-5) they and the
//       temperatures T = tb x are within 2^-21 (1 + 2^-35) of one another, half of what dm_div3's seeds tolerate (2^-20);
//   (2) the same for the compressibilities: |dZ| / Z <= 2 |dZ| with |dZ| <= d_pt ptmax amax + ptmax |lapse| eps (|a1| + 2 tm a2)
//       + 2 d_pt ptmax^2 d, where d_pt = (g0 M / R + |lapse|) eps / Tmin bounds the relative change of p / T over eps
//       (p is a power or an exponential of h with logarithmic derivative g0 M / (R T));
//   (3) |1 - Z| <= ptmax amax + ptmax^2 d <= 2^-10.5 and n - 1 = k_refr (p/T) / Z <= k_refr ptmax / (1 - 2^-10.5) <= 2^-10.5: the
//       squares are the seed errors of 2 - Z for 1/Z and of 1 - (n - 1) for 1/n, 2^-21 — half of what the seeded divisions are
//       tested for (2^-20).  (Standard air at sea level: |1 - Z| = 4.1e-4, n - 1 = 2.8e-4; 2^-10.5 = 6.9e-4.)
// Every left side is evaluated with a margin of 1 % for the rounding of the bound itself.
inline bool atm_interval_tight(const AtmTable& t, int k, double lo, double hi) {
  const double a0 = 1.58123e-6, a1 = -2.9331e-8, a2 = 1.1043e-10, d = 1.83e-11, eps = 0.01, gmr = 9.80665 * 0.0289644 / 8.31432;
  const double two21 = 4.76837158203125e-07, seed = 6.9053396600248786e-04; // 2^-21, 2^-10.5
  if (t.seg(k).cubic || !(lo < hi)) return false;
  const double al = dm_fabs(t.seg(k).lapse);
  // In pieces: over a Linear segment T is monotone in h and so is p / T (a power of T, or an exponential of h), so the end points of
  // a piece bound both; a(c) = a0 + c (a1 + c a2) is a convex parabola in c = T - 273.15, so |a| is largest at an end point or at
  // its vertex.  The quantities are smooth: 64 pieces leave the bounds within a few per cent of the true maxima.
  const int pieces = 64;
  const double cv = -a1 / (2.0 * a2), av = dm_fabs(a0 + cv * (a1 + cv * a2)); // the vertex of a(c) and |a| there
  double h0 = lo, t0 = atm_seg_temperature(t, k, lo), pt0 = t.seg(k).pb * atm_pressure_ratio(t, k, lo) / t0;
  for (int i = 1; i <= pieces; i++) {
    const double h1 = i == pieces ? hi : lo + (hi - lo) * (double)i / (double)pieces;
    const double t1 = atm_seg_temperature(t, k, h1), pt1 = t.seg(k).pb * atm_pressure_ratio(t, k, h1) / t1;
    const double tmin = t0 < t1 ? t0 : t1, tmax = t0 < t1 ? t1 : t0;
    if (!(tmin >= 1.0) || !(pt0 > 0.0) || !(pt1 > 0.0)) return false;
    const double ptmax = (pt0 > pt1 ? pt0 : pt1) * 1.000001;
    const double c0 = t0 - 273.15, c1 = t1 - 273.15, cmin = tmin - 273.15, cmax = tmax - 273.15;
    const double e0 = dm_fabs(a0 + c0 * (a1 + c0 * a2)), e1 = dm_fabs(a0 + c1 * (a1 + c1 * a2));
    double amax = e0 > e1 ? e0 : e1;
    if (cmin <= cv && cv <= cmax && av > amax) amax = av;
    const double tm = dm_fabs(cmin) > dm_fabs(cmax) ? dm_fabs(cmin) : dm_fabs(cmax);
    if (!(1.01 * al * eps / tmin <= two21)) return false;                                          // (1)
    const double d_pt = 1.01 * (gmr + al) * eps / tmin;
    const double dz = d_pt * ptmax * amax + ptmax * al * eps * (dm_fabs(a1) + 2.0 * tm * a2) + 2.0 * d_pt * ptmax * ptmax * d;
    if (!(1.01 * 2.0 * dz <= two21)) return false;                                                 // (2)
    if (!(1.01 * (ptmax * amax + ptmax * ptmax * d) <= seed)) return false;                        // (3) |1 - Z|
    if (!(1.01 * t.k_refr * ptmax / (1.0 - seed) <= seed)) return false;                           // (3) n - 1 = k (p/T) / Z, Z >= 1 - seed
    h0 = h1, t0 = t1, pt0 = pt1;
  }
  (void)h0;
  return true;
}

// Fills safe_lo / safe_hi / alt_lo / alt_hi: for every segment an interval around an anchor altitude (sea level, the
// segment's base, its middle or an end) that atm_interval_certified accepts, inside the global band of altitudes [alt_lo, alt_hi] =
// [max(-100 km, 1 km - radius), 10 000 km].  Nothing is certified for a step outside 1 mm .. 1e8 m, a radius outside 1 km .. 1e12 m
// or a refractivity constant outside 1e-12 .. 1e-3 (a wavelength on a resonance of the dispersion formula).
inline void atm_certify(AtmTable& t, bool spherical, double radius, double step) {
  for (int k = 0; k < t.n; k++) {
    t.seg(k).safe_lo = t.seg(k).tight_lo = dm_inf();
    t.seg(k).safe_hi = t.seg(k).tight_hi = -dm_inf();
    t.seg(k).flags = !t.seg(k).cubic && t.seg(k).lapse == 0.0 ? ATM_SEG_ISOTHERMAL : 0;
    t.seg(k).k_refr = t.k_refr;
    // the exponents of the three points of one right-hand side: e = expo1 log(x), log arguments within 2^-21 of one another on
    // a tight segment (+ the rounding of log itself: 4 ulp of a value below 12), or e = expo1 (h - hb) with h 1 cm apart
    const double ae = dm_fabs(t.seg(k).expo1);
    const double de = t.seg(k).lapse != 0.0 ? ae * (4.76837158203125e-07 + 1.0e-14) : ae * 0.01 * (1.0 + 1.0e-9);
    const double margin = 1.01 * de * DM_INVLN2N + 1.0e-6;
    t.seg(k).exp_thr = margin < 0.5 ? 0.5 - margin : -1.0; // (NaN: -1)
    if (!(t.seg(k).exp_thr >= 0.0)) t.seg(k).exp_thr = -1.0;
  }
  t.alt_lo = dm_inf();
  t.alt_hi = -dm_inf();
  if (!(step >= 1.0e-3 && step <= 1.0e8)) return;
  if (spherical && !(radius >= 1.0e3 && radius <= 1.0e12)) return;
  if (!(t.k_refr >= 1.0e-12 && t.k_refr <= 1.0e-3)) return;
  double gl = -1.0e5;
  const double gh = 1.0e7;
  if (spherical && 1000.0 - radius > gl) gl = 1000.0 - radius;
  t.alt_lo = gl;
  t.alt_hi = gh;
  for (int k = 0; k < t.n; k++) {
    const double L = k > 0 && t.seg(k).from > gl ? t.seg(k).from : gl;
    const double H = k + 1 < t.n && t.seg(k + 1).from < gh ? t.seg(k + 1).from : gh;
    if (!(L < H)) continue;
    double lo = L, hi = H;
    if (!atm_interval_certified(t, k, lo, hi)) {
      const double cands[5] = {0.0, t.seg(k).hb, 0.5 * (L + H), L, H};
      double a = 0.0;
      bool have = false;
      for (int i = 0; i < 5 && !have; i++) {
        a = cands[i] < L ? L : (cands[i] > H ? H : cands[i]);
        have = atm_interval_certified(t, k, a, a);
      }
      if (!have) continue;
      // upwards first (that is where rays spend their time), then downwards with the upper end fixed: certified jointly
      if (atm_interval_certified(t, k, a, H)) {
        hi = H;
      } else {
        double good = a, bad = H;
        for (int it = 0; it < 60; it++) {
          const double mid = 0.5 * (good + bad);
          if (atm_interval_certified(t, k, a, mid)) good = mid;
          else bad = mid;
        }
        hi = good;
      }
      if (atm_interval_certified(t, k, L, hi)) {
        lo = L;
      } else {
        double good = a, bad = L;
        for (int it = 0; it < 60; it++) {
          const double mid = 0.5 * (good + bad);
          if (atm_interval_certified(t, k, mid, hi)) good = mid;
          else bad = mid;
        }
        lo = good;
      }
      if (!atm_interval_certified(t, k, lo, hi) || !(lo < hi)) continue;
    }
    t.seg(k).safe_lo = lo;
    t.seg(k).safe_hi = hi;
    // The tight part of the certified interval: all of it, or — the lowest layer of a physical atmosphere is certified tens of
    // kilometres further down than it is tight, the highest up to where its extrapolated temperature reaches 1 K and tight while it
    // is above ~40 K — from -1500 m (or the interval's lower end) upwards as far as the bounds hold, found by bisection.  Rays
    // march between -1000 m (rectilinear.rs:178) and a few tens of kilometres; a lane outside the tight part costs its wavefront
    // the votes of dm_div3 and three v_rcp_f64 for that stage, nothing else.
    const double cut_lo = lo > -1500.0 ? lo : -1500.0;
    if (atm_interval_tight(t, k, lo, hi)) {
      t.seg(k).tight_lo = lo, t.seg(k).tight_hi = hi;
      t.seg(k).flags |= ATM_SEG_TIGHT;
    } else if (cut_lo < hi && atm_interval_tight(t, k, cut_lo, cut_lo + 1.0 < hi ? cut_lo + 1.0 : hi)) {
      const double tl = atm_interval_tight(t, k, lo, cut_lo + 1.0 < hi ? cut_lo + 1.0 : hi) ? lo : cut_lo;
      double good = cut_lo + 1.0 < hi ? cut_lo + 1.0 : hi, bad = hi;
      if (atm_interval_tight(t, k, tl, hi)) {
        good = hi;
      } else {
        for (int it = 0; it < 48; it++) {
          const double mid = 0.5 * (good + bad);
          if (atm_interval_tight(t, k, tl, mid)) good = mid;
          else bad = mid;
        }
      }
      t.seg(k).tight_lo = tl, t.seg(k).tight_hi = good;
      t.seg(k).flags |= ATM_SEG_TIGHT;
    }
  }
}

ATMRT_HD double atm_temperature(const AtmTable& a, double h) { return atm_seg_temperature(a, atm_layer(a, h), h); }
ATMRT_HD double atm_pressure(const AtmTable& a, double h) {
  int k = atm_layer(a, h);
  return a.seg(k).pb * atm_pressure_ratio(a, k, h);
}

// FAST selects the GPU's shortcut sequences (dm_div ...: the IEEE result for in-range operands only) and is passed as true only for
// evaluations inside a certified altitude interval (AtmTable::safe_lo); FAST = false is the IEEE operation.  Same value on the host.
template <bool FAST>
ATMRT_HD double div_sel(double a, double b) {
  return FAST ? dm_div(a, b) : a / b;
}
// a wave vote on the GPU (the branch it guards then is uniform), the predicate itself on the host
ATMRT_HD bool wave_all(bool p) {
#if defined(__HIP_DEVICE_COMPILE__)
  return __all(p);
#else
  return p;
#endif
}

constexpr double REFR_INV_2EPS = 50.0; // 0.5 / eps of the central difference dn/dh (eps = 0.01 m; the quotient is 50 exactly)
// Ciddor's compressibility of dry air, Z = 1 - pt (a0 + a1 t + a2 t^2) + pt^2 d with pt = p / T and t = T - 273.15, as
// oracle/atmosphere.c evaluates it: fma(pt, fma(pt, d, -A), 1), A = fma(t, fma(t, a2, a1), a0).  -A is formed directly (every
// constant negated: the same magnitude bit for bit, rounding to nearest is symmetric), so no negation is issued.
ATMRT_HD double ciddor_z(double pt, double t) {
  const double a0 = 1.58123e-6, a1 = -2.9331e-8, a2 = 1.1043e-10, d = 1.83e-11;
  const double na = DM_FMA_VVS(t, DM_FMA_VSV(t, -a2, -a1), -a0);
  return DM_FMA(pt, DM_FMA_VSV(pt, d, na), 1.0);
}
// Ciddor's (n - 1) = K (p/T) / Z for dry air, from the density term pt = p / T and t = T - 273.15
template <bool FAST = false>
ATMRT_HD double refr_from_pt(double k_refr, double pt, double t) {
  return 1.0 + div_sel<FAST>(k_refr * pt, ciddor_z(pt, t));
}
template <bool FAST = false>
ATMRT_HD double refr_from_tp(double k_refr, double temp, double p) { // Spline segments: p and T come separately
  return refr_from_pt<FAST>(k_refr, div_sel<FAST>(p, temp), temp - 273.15);
}

// refr_from_pt at the three points of one ODE right-hand side (same layer, heights 1 cm apart): the same values as three calls; the
// division site shares its reciprocal refinement across the points (dm_div3)
ATMRT_HD void refr_from_pt3(double k_refr, double pt0, double pt1, double pt2, double t0, double t1, double t2, double& n0, double& n1,
                            double& n2) {
  const double z0 = ciddor_z(pt0, t0), z1 = ciddor_z(pt1, t1), z2 = ciddor_z(pt2, t2);
  double q0, q1, q2;
  dm_div3(k_refr * pt0, z0, k_refr * pt1, z1, k_refr * pt2, z2, &q0, &q1, &q2);
  n0 = 1.0 + q0;
  n1 = 1.0 + q1;
  n2 = 1.0 + q2;
}

// the same on a TIGHT segment (AtmSeg::flags): no vote, the reciprocal of Z seeded by 2 - Z; q0 = n0 - 1 seeds the caller's 1 / n0
ATMRT_HD void refr_from_pt3_tight(double k_refr, double pt0, double pt1, double pt2, double t0, double t1, double t2, double& n0, double& n1,
                                  double& n2, double& q0) {
  const double z0 = ciddor_z(pt0, t0), z1 = ciddor_z(pt1, t1), z2 = ciddor_z(pt2, t2);
  double q1, q2;
  dm_div3_seeded(k_refr * pt0, z0, k_refr * pt1, z1, k_refr * pt2, z2, 1, 2.0 - z0, &q0, &q1, &q2);
  n0 = 1.0 + q0;
  n1 = 1.0 + q1;
  n2 = 1.0 + q2;
}

// a knot interval of a Spline temperature function
template <bool FAST = false>
ATMRT_HD double refr_n_cubic_segment(double k_refr, double hb, double tb, double pb, double c1, double c2, double c3, double expo, double h) {
  double temp = seg_temperature(tb, c1, c2, c3, h - hb);
  double p = pb * dm_exp(expo * seg_inv_t_integral(tb, c1, c2, c3, h - hb));
  return refr_from_tp<FAST>(k_refr, temp, p);
}

// Environment::n(h) for a point known to lie in the segment with these parameters.  CUBIC = false instantiates only the
// closed-form path of Linear functions: the stepping kernels are compiled in both variants and the host picks by
// whether the atmosphere has Spline segments (inlining the quadrature path twelve times per RK4 step costs 9 % on US-76).
template <bool CUBIC = true, bool FAST = false>
ATMRT_HD double refr_n_layer(double k_refr, const AtmSeg& sg, double h) {
  if (CUBIC && sg.cubic) return refr_n_cubic_segment<FAST>(k_refr, sg.hb, sg.tb, sg.pb, sg.lapse, sg.c2, sg.c3, sg.expo, h);
  const double dh = h - sg.hb;
  const double x = DM_FMA(sg.gtb, dh, 1.0);
  const double pt = sg.lapse != 0.0 ? sg.ptb * dm_pow(x, sg.expo1) : sg.ptb * dm_exp(sg.expo1 * dh);
  return refr_from_pt<FAST>(k_refr, pt, DM_FMA(sg.tb, x, -273.15));
}

// refr_n_layer at three points of one layer.  Same values as three calls; on the GPU the range guards of log and exp are
// taken once per wavefront (a vote), so that the three pow evaluations form one basic block and their table look-ups overlap.
ATMRT_HD void pow3(double x0, double x1, double x2, double y, double& r0, double& r1, double& r2) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (__all(dm_log_in_main_range(x0) && dm_log_in_main_range(x1) && dm_log_in_main_range(x2))) {
    const double e0 = y * dm_log_core_pow(x0), e1 = y * dm_log_core_pow(x1), e2 = y * dm_log_core_pow(x2);
    if (__all(dm_exp_in_main_range(e0) && dm_exp_in_main_range(e1) && dm_exp_in_main_range(e2))) {
      r0 = dm_exp_main(e0);
      r1 = dm_exp_main(e1);
      r2 = dm_exp_main(e2);
      return;
    }
    r0 = dm_exp(e0);
    r1 = dm_exp(e1);
    r2 = dm_exp(e2);
    return;
  }
#endif
  r0 = dm_pow(x0, y);
  r1 = dm_pow(x1, y);
  r2 = dm_pow(x2, y);
}
ATMRT_HD void exp3(double e0, double e1, double e2, double& r0, double& r1, double& r2) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (__all(dm_exp_in_main_range(e0) && dm_exp_in_main_range(e1) && dm_exp_in_main_range(e2))) {
    r0 = dm_exp_main(e0);
    r1 = dm_exp_main(e1);
    r2 = dm_exp_main(e2);
    return;
  }
#endif
  r0 = dm_exp(e0);
  r1 = dm_exp(e1);
  r2 = dm_exp(e2);
}
// the same for arguments the certificate has already placed in the main ranges of log and exp (atm_interval_certified): no votes
ATMRT_HD void pow3_in_range(double x0, double x1, double x2, double y, double& r0, double& r1, double& r2) {
  const double e0 = y * dm_log_core_pow(x0), e1 = y * dm_log_core_pow(x1), e2 = y * dm_log_core_pow(x2);
  r0 = dm_exp_main(e0);
  r1 = dm_exp_main(e1);
  r2 = dm_exp_main(e2);
}
// the same on a TIGHT segment: the shared-row forms of detmath.h where no lane sits on a table edge (wave votes), else the plain ones
ATMRT_HD void pow3_tight(double x0, double x1, double x2, double y, double thr, double& r0, double& r1, double& r2) {
#if defined(__HIP_DEVICE_COMPILE__)
  double l0, l1, l2;
  if (!dm_log3_core_pow_shared(x0, x1, x2, &l0, &l1, &l2)) {
    l0 = dm_log_core_pow(x0);
    l1 = dm_log_core_pow(x1);
    l2 = dm_log_core_pow(x2);
  }
  const double e0 = y * l0, e1 = y * l1, e2 = y * l2;
  if (!dm_exp3_main_shared(e0, e1, e2, thr, &r0, &r1, &r2)) {
    r0 = dm_exp_main(e0);
    r1 = dm_exp_main(e1);
    r2 = dm_exp_main(e2);
  }
#else
  (void)thr;
  pow3_in_range(x0, x1, x2, y, r0, r1, r2);
#endif
}
ATMRT_HD void exp3_tight(double e0, double e1, double e2, double thr, double& r0, double& r1, double& r2) {
#if defined(__HIP_DEVICE_COMPILE__)
  if (dm_exp3_main_shared(e0, e1, e2, thr, &r0, &r1, &r2)) return;
#else
  (void)thr;
#endif
  r0 = dm_exp_main(e0);
  r1 = dm_exp_main(e1);
  r2 = dm_exp_main(e2);
}
// Only for points inside a certified interval [safe_lo, safe_hi) of a segment (the callers' wave votes): there T / tb is a positive
// normal number and |expo log(T / tb)| (|expo (h - hb)| on an isothermal segment) is at most 645, so log and exp take their main
// branches unasked — the values of refr_n_layer at the three points.
template <bool CUBIC = true>
ATMRT_HD void refr_n_layer3(double k_refr, int cubic, int flags, double exp_thr, double hb, double tb, double gtb, double ptb, double expo1,
                            double pb, double lapse, double c2, double c3, double expo, double h0, double h1, double h2, double& n0,
                            double& n1, double& n2, double& q0) {
  if (CUBIC && cubic) {
    n0 = refr_n_cubic_segment<true>(k_refr, hb, tb, pb, lapse, c2, c3, expo, h0);
    n1 = refr_n_cubic_segment<true>(k_refr, hb, tb, pb, lapse, c2, c3, expo, h1);
    n2 = refr_n_cubic_segment<true>(k_refr, hb, tb, pb, lapse, c2, c3, expo, h2);
    return;
  }
  const double d0 = h0 - hb, d1 = h1 - hb, d2 = h2 - hb;
  const double x0 = DM_FMA(gtb, d0, 1.0), x1 = DM_FMA(gtb, d1, 1.0), x2 = DM_FMA(gtb, d2, 1.0);
  const double t0 = DM_FMA(tb, x0, -273.15), t1 = DM_FMA(tb, x1, -273.15), t2 = DM_FMA(tb, x2, -273.15);
  double r0, r1, r2;
  if (flags & ATM_SEG_TIGHT) {
    if (!(flags & ATM_SEG_ISOTHERMAL)) pow3_tight(x0, x1, x2, expo1, exp_thr, r0, r1, r2);
    else exp3_tight(expo1 * d0, expo1 * d1, expo1 * d2, exp_thr, r0, r1, r2);
    refr_from_pt3_tight(k_refr, ptb * r0, ptb * r1, ptb * r2, t0, t1, t2, n0, n1, n2, q0);
    return;
  }
  if (!(flags & ATM_SEG_ISOTHERMAL)) {
    pow3_in_range(x0, x1, x2, expo1, r0, r1, r2);
  } else {
    r0 = dm_exp_main(expo1 * d0);
    r1 = dm_exp_main(expo1 * d1);
    r2 = dm_exp_main(expo1 * d2);
  }
  refr_from_pt3(k_refr, ptb * r0, ptb * r1, ptb * r2, t0, t1, t2, n0, n1, n2);
}

// Environment::n(h) (renderer/mod.rs:425 is the only direct call site; the stepper uses it too).  IEEE operations throughout: the
// generic evaluation, valid for any atmosphere at any altitude.
ATMRT_HD double refr_n(const AtmTable& a, double h) {
  int k = atm_layer(a, h);
  return refr_n_layer(a.k_refr, a.seg(k), h);
}
ATMRT_HD double refr_dn(const AtmTable& a, double h) {
  const double eps = 0.01;
  double n1 = refr_n(a, h - eps);
  double n2 = refr_n(a, h + eps);
  return (n2 - n1) * REFR_INV_2EPS;
}

// Environment::n(h) for a point INSIDE the certified part of its segment: shortcut divisions, and log / exp by their main branches
// without range tests (atm_interval_certified bounds T / tb to a positive normal number and the exponent to |e| <= 645).
// Straight-line code.  For a point outside, the result is garbage of no consequence (every table index is masked): callers
// discard it (refr_n_speculative's `certified` flag).
template <bool CUBIC>
ATMRT_HD double refr_n_layer_inrange(double k_refr, int cubic, double hb, double tb, double gtb, double ptb, double expo1, double pb,
                                     double lapse, double c2, double c3, double expo, double h) {
  if (CUBIC && cubic) {
    const double temp = seg_temperature(tb, lapse, c2, c3, h - hb);
    const double p = pb * dm_exp_main(expo * seg_inv_t_integral(tb, lapse, c2, c3, h - hb));
    return refr_from_tp<true>(k_refr, temp, p);
  }
  const double dh = h - hb;
  const double x = DM_FMA(gtb, dh, 1.0);
  const double e = lapse != 0.0 ? expo1 * dm_log_core_pow(x) : expo1 * dh;
  return refr_from_pt<true>(k_refr, ptb * dm_exp_main(e), DM_FMA(tb, x, -273.15));
}

// One evaluation of n(h) with the shortcut divisions, organised for the wavefront of the path kernel (atmrt_paths.hip): `hint` is the
// layer of the lane's previous evaluation.  When every active lane is inside the certified part of the first lane's layer (rays
// below 11 km in a physical atmosphere) the layer search is two compares and the layer parameters are wave-uniform scalars —
// kept in `cache` (SGPRs) from one evaluation to the next and re-read only when the hinted layer changes, so that no scalar-load
// round trip sits on the kernel's dependent chain; otherwise every lane makes the full search and gathers its own parameters
// (eight rays of different elevations share a wavefront there: the normal case above 11 km).  `certified` (per lane): the point
// lies in the certified part of its layer, i.e. the value is refr_n's.  If it is false the value is NOT to be used — the caller
// verifies the flags of a whole step with one vote and repeats the step with IEEE operations when one is false.  GPU only.
#if defined(__HIPCC__)
struct AtmLayerCache {
  int k = -1;
  int cubic = 0;
  double safe_lo = 0.0, safe_hi = 0.0, hb = 0.0, tb = 0.0, gtb = 0.0, ptb = 0.0, expo1 = 0.0, pb = 0.0, lapse = 0.0, c2 = 0.0, c3 = 0.0, expo = 0.0, k_refr = 0.0;
};
template <bool CUBIC>
__device__ __forceinline__ double refr_n_speculative(const AtmTable& a, AtmLayerCache& cache, double h, int& hint, bool& certified) {
  const int ku = __builtin_amdgcn_readfirstlane(hint);
  if (ku != cache.k) { // wave-uniform
    const AtmConstSeg ks = atm_const_seg(a, ku);
    cache.k = ku;
    cache.cubic = ks->cubic;
    cache.safe_lo = ks->safe_lo, cache.safe_hi = ks->safe_hi, cache.hb = ks->hb, cache.tb = ks->tb, cache.gtb = ks->gtb, cache.ptb = ks->ptb, cache.expo1 = ks->expo1, cache.pb = ks->pb;
    cache.lapse = ks->lapse, cache.c2 = ks->c2, cache.c3 = ks->c3, cache.expo = ks->expo;
    cache.k_refr = atm_const_table(a)->k_refr;
  }
  if (__all(h >= cache.safe_lo && h < cache.safe_hi)) {
    certified = true;
    return refr_n_layer_inrange<CUBIC>(cache.k_refr, cache.cubic, cache.hb, cache.tb, cache.gtb, cache.ptb, cache.expo1, cache.pb, cache.lapse, cache.c2, cache.c3, cache.expo, h);
  }
  const int k = atm_layer(a, h);
  hint = k;
  const AtmSeg& sg = a.seg(k);
  certified = h >= sg.safe_lo && h < sg.safe_hi;
  return refr_n_layer_inrange<CUBIC>(a.k_refr, sg.cubic, sg.hb, sg.tb, sg.gtb, sg.ptb, sg.expo1, sg.pb, sg.lapse, sg.c2, sg.c3, sg.expo, h);
}
#endif

// n(h) and dn/dh(h) of one ODE right-hand side: the three evaluations at h, h - eps and h + eps.  When all three points of
// every active lane lie in the certified part of the hinted layer (they are 1 cm apart), one check and one set of scalar layer
// parameters serve the three evaluations and their instruction streams interleave (the table look-ups of exp/log are the long
// latencies here).  Returns whether that path was taken (wave-uniform): n then is in [1, 2^9] and the caller may go on with the shortcuts.
template <bool CUBIC>
ATMRT_HD bool refr_n_dn_hint(const AtmTable& a, double h, int& hint, double& n, double& dn) {
#if defined(__HIP_DEVICE_COMPILE__)
  const double eps = 0.01;
  const int ku = __builtin_amdgcn_readfirstlane(hint);
  const double h1 = h - eps, h2 = h + eps;
  // the table is read-only for the whole launch: reading it through the constant address space makes these scalar loads
  const AtmConstSeg ks = atm_const_seg(a, ku);
  const bool tight = __all(h1 >= ks->tight_lo && h2 < ks->tight_hi);
  if (tight || __all(h1 >= ks->safe_lo && h2 < ks->safe_hi)) {
    const double k_refr = ks->k_refr, hb = ks->hb, tb = ks->tb, gtb = ks->gtb, ptb = ks->ptb, expo1 = ks->expo1, pb = ks->pb, lapse = ks->lapse, c2 = ks->c2, c3 = ks->c3, expo = ks->expo;
    const int cubic = ks->cubic, flags = tight ? ks->flags : ks->flags & ~ATM_SEG_TIGHT;
    const double exp_thr = ks->exp_thr;
    double n1, n2, q0;
    refr_n_layer3<CUBIC>(k_refr, cubic, flags, exp_thr, hb, tb, gtb, ptb, expo1, pb, lapse, c2, c3, expo, h, h1, h2, n, n1, n2, q0);
    dn = (n2 - n1) * REFR_INV_2EPS;
    return true;
  }
  // generic: per-lane layer search, IEEE operations.  The hint follows the lane, so that the fast path resumes once the
  // wavefront is back inside one certified interval.
  {
    const int k = atm_layer(a, h);
    hint = k;
    n = refr_n_layer<CUBIC, false>(a.k_refr, a.seg(k), h);
    const int k1 = atm_layer(a, h1), k2 = atm_layer(a, h2);
    const double n1 = refr_n_layer<CUBIC, false>(a.k_refr, a.seg(k1), h1);
    const double n2 = refr_n_layer<CUBIC, false>(a.k_refr, a.seg(k2), h2);
    dn = (n2 - n1) * REFR_INV_2EPS;
    return false;
  }
#else
  (void)hint;
  n = refr_n(a, h);
  dn = refr_dn(a, h);
  return false;
#endif
}

// ---------------------------------------------------------------------------------------------
// Earth model + directional calculators (src/utils/earth_model/{mod,directional_calc}.rs)
// ---------------------------------------------------------------------------------------------

constexpr double DEGREE_DISTANCE = 10000000.0 / 90.0; // earth_model/mod.rs:12
constexpr double EARTH_R = 6371000.0;                 // :14
constexpr double WGS84_A = 6378137.0;                 // :15
constexpr double WGS84_B = 6356752.314245;            // :16

// EarthModel resolved once on the host into the four behaviours the path needs.
struct Earth {
  int32_t calc;       // 0 AzEq, 1 FlDs, 2 Spherical, 3 Ellipsoid     (coords_at_dist_calc, mod.rs:114-145)
  int32_t flat_dirs;  // bit 0: world_directions / as_cartesian flat family  (mod.rs:31-57,59-93); bit 1: EARTH_FAST_DIV
  int32_t cart;       // 0 flat, 1 spherical(cart_radius), 2 ellipsoid(a,b)
  int32_t spherical;  // EarthShape::Spherical{shape_radius} vs Flat  (to_shape, mod.rs:95-112)
  double calc_radius; // SphericalCalc radius
  double cart_radius;
  double a, b;
  double shape_radius;
};
// bit 1 of Earth::flat_dirs: x / calc_radius may take dm_div — the radius and every stepper distance x lie within 1e-30 .. 1e30
// (atmrt_set_params).  A bit of an existing field rather than a new one keeps the struct at 56 bytes, which the out-of-line object
// code (close_mask_impl, atmrt_march_impl.h) receives in registers.  Size is NOT a correctness matter: round 2 saw the tracer lose
// its objects with a 64-byte Earth, but that was the IPRA failure of profiles/r03/ipra/README.md (a 64-byte Earth passes the object
// tests once the calling units are built with -enable-ipra=0, checked in round 3).
constexpr int32_t EARTH_FLAT_DIRS = 1, EARTH_FAST_DIV = 2;

ATMRT_HD int earth_resolve(const atmrt_earth_model_t& m, Earth& e) {
  e.calc_radius = e.cart_radius = e.a = e.b = e.shape_radius = 0.0;
  switch (m.kind) {
    case ATMRT_EARTH_SIMPLE_SPHERE:
      e.calc = 2; e.flat_dirs = 0; e.cart = 1; e.spherical = 1;
      e.calc_radius = e.cart_radius = e.shape_radius = EARTH_R;
      return 0;
    case ATMRT_EARTH_SPHERICAL:
      e.calc = 2; e.flat_dirs = 0; e.cart = 1; e.spherical = 1;
      e.calc_radius = e.cart_radius = e.shape_radius = m.radius;
      return 0;
    case ATMRT_EARTH_ELLIPSOID:
    case ATMRT_EARTH_WGS84:
      e.calc = 3; e.flat_dirs = 0; e.cart = 2; e.spherical = 1;
      e.a = m.kind == ATMRT_EARTH_WGS84 ? WGS84_A : m.a;
      e.b = m.kind == ATMRT_EARTH_WGS84 ? WGS84_B : m.b;
      e.shape_radius = (2.0 * e.a + e.b) / 3.0;
      return 0;
    case ATMRT_EARTH_AZIMUTHAL_EQUIDISTANT:
      e.calc = 0; e.flat_dirs = 1; e.cart = 0; e.spherical = 0;
      return 0;
    case ATMRT_EARTH_FLAT_DISTORTED:
      e.calc = 1; e.flat_dirs = 1; e.cart = 0; e.spherical = 0;
      return 0;
    case ATMRT_EARTH_OBSERVER_AE:
    case ATMRT_EARTH_SIMPLE_OBSERVER_AE:
      e.calc = 2; e.flat_dirs = 1; e.cart = 0; e.spherical = 0;
      e.calc_radius = m.kind == ATMRT_EARTH_OBSERVER_AE ? m.radius : EARTH_R;
      return 0;
    default: return -1;
  }
}

// spherical_directions, mod.rs:155-172
ATMRT_HD void spherical_directions(double lat, double lon, Vec3& dirn, Vec3& dire, Vec3& dirup) {
  double sinlon, coslon, sinlat, coslat;
  dm_sincos(dm_to_radians(lon), &sinlon, &coslon);
  dm_sincos(dm_to_radians(lat), &sinlat, &coslat);
  dirup = v3(coslat * coslon, coslat * sinlon, sinlat);
  dirn = v3(-sinlat * coslon, -sinlat * sinlon, coslat);
  dire = v3(-sinlon, coslon, 0.0);
}

// EarthModel::world_directions, mod.rs:31-57
ATMRT_HD void world_directions(const Earth& e, double lat, double lon, Vec3& n, Vec3& ea, Vec3& up) {
  if (e.flat_dirs & EARTH_FLAT_DIRS) {
    double sinlon, coslon;
    dm_sincos(dm_to_radians(lon), &sinlon, &coslon);
    n = v3(-coslon, -sinlon, 0.0);
    ea = v3(-sinlon, coslon, 0.0);
    up = v3(0.0, 0.0, 1.0);
  } else {
    spherical_directions(lat, lon, n, ea, up);
  }
}

// EarthModel::as_cartesian, mod.rs:59-93 (+ spherical_to_cartesian :148-153)
ATMRT_HD Vec3 as_cartesian(const Earth& e, double lat, double lon, double elev) {
  if (e.cart == 1) {
    double r = e.cart_radius + elev;
    double sl, cl, so, co;
    dm_sincos(dm_to_radians(lat), &sl, &cl);
    dm_sincos(dm_to_radians(lon), &so, &co);
    return v3(r * cl * co, r * cl * so, r * sl);
  }
  if (e.cart == 2) {
    double e2 = 1.0 - (e.b * e.b) / (e.a * e.a);
    double sl, cl, so, co;
    dm_sincos(dm_to_radians(lat), &sl, &cl);
    dm_sincos(dm_to_radians(lon), &so, &co);
    double n = e.a / dm_sqrt(1.0 - e2 * (sl * sl));
    return v3((n + elev) * cl * co, (n + elev) * cl * so, (n * (1.0 - e2) + elev) * sl);
  }
  double r = (90.0 - lat) * DEGREE_DISTANCE;
  double so, co;
  dm_sincos(dm_to_radians(lon), &so, &co);
  return v3(r * co, r * so, elev);
}

// Box<dyn DirectionalCalc> flattened: one POD for the four implementors.
struct DirCalc {
  // Spherical: pos = unit up vector at start, dir = unit tangent.  AzEq: pos cartesian, dir = dir_v.
  Vec3 pos, dir;
  // FlDs: p0 = start lat, p1 = start lon, p2 = cos(dir), p3 = sin(dir), p4 = cos(start lat)
  // Ellipsoid: p0..p9 = b, f, red_lat, lon, az1, alfa, sig1, cap_a, cap_b, cap_c
  double p[10];
};

// EarthModel::coords_at_dist_calc, mod.rs:114-145 and the ::new of each calculator
ATMRT_HD void dircalc_new(const Earth& e, double lat, double lon, double dir, DirCalc& c) {
  for (int i = 0; i < 10; i++) c.p[i] = 0.0;
  c.pos = c.dir = v3(0.0, 0.0, 0.0);
  if (e.calc == 2) { // SphericalCalc::new, directional_calc.rs:56-69
    Vec3 dirn, dire, pos;
    spherical_directions(lat, lon, dirn, dire, pos);
    double sindir, cosdir;
    dm_sincos(dm_to_radians(dir), &sindir, &cosdir);
    c.pos = pos;
    c.dir = dirn * cosdir + dire * sindir;
  } else if (e.calc == 0) { // AzEqCalc, mod.rs:116-125
    Vec3 vn, ve, vu;
    c.pos = as_cartesian(e, lat, lon, 0.0);
    world_directions(e, lat, lon, vn, ve, vu);
    double s, co;
    dm_sincos(dm_to_radians(dir), &s, &co);
    c.dir = vn * co + ve * s;
  } else if (e.calc == 1) { // FlDsCalc::new, directional_calc.rs:35-39
    c.p[0] = lat;
    c.p[1] = lon;
    dm_sincos(dm_to_radians(dir), &c.p[3], &c.p[2]);
    c.p[4] = dm_cos(dm_to_radians(lat));
  } else { // EllipsoidCalc::new, directional_calc.rs:103-131
    double a = e.a, b = e.b;
    double la = dm_to_radians(lat), lo = dm_to_radians(lon), az1 = dm_to_radians(dir);
    double f = (a - b) / a;
    double red_lat = dm_atan((1.0 - f) * dm_tan(la));
    double sig1 = dm_atan(dm_tan(red_lat) / dm_cos(az1));
    double alfa = dm_asin(dm_cos(red_lat) * dm_sin(az1));
    double ca = dm_cos(alfa);
    double u2 = ca * ca * (a * a - b * b) / (b * b);
    c.p[0] = b;
    c.p[1] = f;
    c.p[2] = red_lat;
    c.p[3] = lo;
    c.p[4] = az1;
    c.p[5] = alfa;
    c.p[6] = sig1;
    c.p[7] = 1.0 + u2 / 256.0 * (64.0 + u2 * (-12.0 + 5.0 * u2));
    c.p[8] = u2 / 512.0 * (128.0 + u2 * (-64.0 + 37.0 * u2));
    c.p[9] = f / 16.0 * (ca * ca) * (4.0 + f * (4.0 - 3.0 * (ca * ca)));
  }
}

// DirectionalCalc::coords_at_dist
ATMRT_HD void coords_at_dist(const Earth& e, const DirCalc& c, double dist, double& lat, double& lon) {
  if (e.calc == 2) { // SphericalCalc, directional_calc.rs:72-85
    double ang = (e.flat_dirs & EARTH_FAST_DIV) ? dm_div(dist, e.calc_radius) : dist / e.calc_radius;
    double sinang, cosang;
    dm_sincos(ang, &sinang, &cosang);
    double fx = c.pos.x * cosang + c.dir.x * sinang;
    double fy = c.pos.y * cosang + c.dir.y * sinang;
    double fz = c.pos.z * cosang + c.dir.z * sinang;
    lat = dm_to_degrees(dm_asin(fz));
    lon = dm_to_degrees(dm_atan2(fy, fx));
  } else if (e.calc == 0) { // AzEqCalc, directional_calc.rs:20-28
    double px = c.pos.x + c.dir.x * dist, py = c.pos.y + c.dir.y * dist;
    lon = dm_to_degrees(dm_atan2(py, px));
    double r = dm_sqrt(px * px + py * py);
    lat = 90.0 - r / DEGREE_DISTANCE;
  } else if (e.calc == 1) { // FlDsCalc, directional_calc.rs:41-48
    double d_lat = c.p[2] * dist / DEGREE_DISTANCE;
    double d_lon = c.p[3] * dist / DEGREE_DISTANCE / c.p[4];
    lat = c.p[0] + d_lat;
    lon = c.p[1] + d_lon;
  } else { // EllipsoidCalc (Vincenty direct), directional_calc.rs:135-185
    const double b = c.p[0], f = c.p[1], red_lat = c.p[2], lon0 = c.p[3], az1 = c.p[4], alfa = c.p[5],
                 sig1 = c.p[6], cap_a = c.p[7], cap_b = c.p[8], cap_c = c.p[9];
    double sig = dist / b / cap_a, sigm, cs;
    // bounded (the reference loops unbounded and would spin on NaN); 64 is never reached for finite input
    for (int it = 0; it < 64; it++) {
      sigm = 2.0 * sig1 + sig;
      cs = dm_cos(sigm);
      double dsig = cap_b * dm_sin(sig) * (cs + cap_b / 4.0 * dm_cos(sig) * (-1.0 + 2.0 * (cs * cs)));
      double new_sig = dist / b / cap_a + dsig;
      dsig = dm_fabs(new_sig - sig);
      sig = new_sig;
      if (dsig < 1e-10) break;
    }
    sigm = 2.0 * sig1 + sig;
    double srl, crl, ss, csg, sa, saz, caz;
    dm_sincos(red_lat, &srl, &crl);
    dm_sincos(sig, &ss, &csg);
    dm_sincos(az1, &saz, &caz);
    sa = dm_sin(alfa);
    double t = srl * ss - crl * csg * caz;
    double lat2 = dm_atan((srl * csg + crl * ss * caz) / ((1.0 - f) * dm_sqrt(sa * sa + t * t)));
    double lambda = dm_atan(ss * saz / (crl * csg - srl * ss * caz));
    cs = dm_cos(sigm);
    double dl = lambda - (1.0 - cap_c) * f * sa * (sig + cap_c * ss * (cs + cap_c * csg * (-1.0 + 2.0 * (cs * cs))));
    lat = dm_to_degrees(lat2);
    lon = dm_to_degrees(lon0 + dl);
  }
}

// ---------------------------------------------------------------------------------------------
// Terrain mosaic in HBM + bilinear sampler (src/terrain/mod.rs:120-126, tile.rs:28-30; sampler
// modelled on terrain/geotiff.rs:61-100 because crate dted 0.2 is absent)
// ---------------------------------------------------------------------------------------------

struct TileDesc {
  int64_t offset; // first post of the tile in the mosaic buffer
  int32_t n_lat, n_lon;
};

struct TerrainView {
  const int16_t* posts;     // all tiles back to back, each [n_lat][n_lon] south->north, west->east
  const TileDesc* tiles;
  const int32_t* cell_tile; // [n_cells_lat][n_cells_lon] -> tile slot or -1
  int32_t lat_min, lon_min, n_cells_lat, n_cells_lon;
  // max(highest post, 0) + 1 m: a ray sample above it is above the terrain for certain (the bilinear value is a convex
  // combination of posts up to a few ulps, and 0 m stands for every missing tile), so its lookup cannot change a sign test
  double skip_above;
};

// Rust `f as i16` / `f as usize` (saturating, NaN -> 0)
ATMRT_HD int sat_i16(double f) {
  if (f != f) return 0;
  if (f <= -32768.0) return -32768;
  if (f >= 32767.0) return 32767;
  return (int)f;
}
ATMRT_HD int sat_index(double f) {
  if (f != f || f <= 0.0) return 0;
  if (f >= 2147483647.0) return 2147483647;
  return (int)f;
}

// Terrain::get_elev: returns false for None.  4 int16 posts = 8 B of HBM traffic per lookup.
ATMRT_HD bool terrain_get_elev(const TerrainView& tv, double latitude, double longitude, double& elev) {
  int lat = sat_i16(dm_floor(latitude));
  int lon = sat_i16(dm_floor(longitude));
  int ci = lat - tv.lat_min, cj = lon - tv.lon_min;
  if (ci < 0 || cj < 0 || ci >= tv.n_cells_lat || cj >= tv.n_cells_lon) return false;
  int slot = tv.cell_tile[ci * tv.n_cells_lon + cj];
  if (slot < 0) return false;
  const TileDesc td = tv.tiles[slot];
  double min_lat = (double)lat, min_lon = (double)lon;
  double max_lat = min_lat + 1.0, max_lon = min_lon + 1.0;
  if (latitude < min_lat || latitude > max_lat || longitude < min_lon || longitude > max_lon) return false;
  double flat = (latitude - min_lat) * (double)(td.n_lat - 1);
  double flon = (longitude - min_lon) * (double)(td.n_lon - 1);
  int lat_int = sat_index(flat), lon_int = sat_index(flon);
  double lat_frac = flat - (double)lat_int;
  double lon_frac = flon - (double)lon_int;
  if (lat_int == td.n_lat - 1) { // max-edge case, geotiff.rs:77-85
    lat_int -= 1;
    lat_frac += 1.0;
  }
  if (lon_int == td.n_lon - 1) {
    lon_int -= 1;
    lon_frac += 1.0;
  }
  const int16_t* row0 = tv.posts + td.offset + (int64_t)lat_int * td.n_lon + lon_int;
  const int16_t* row1 = row0 + td.n_lon;
  double e00 = (double)row0[0], e10 = (double)row0[1];
  double e01 = (double)row1[0], e11 = (double)row1[1];
  elev = e00 * (1.0 - lon_frac) * (1.0 - lat_frac) + e01 * (1.0 - lon_frac) * lat_frac +
         e10 * lon_frac * (1.0 - lat_frac) + e11 * lon_frac * lat_frac;
  return true;
}
ATMRT_HD double terrain_elev_or_zero(const TerrainView& tv, double lat, double lon) {
  double e;
  return terrain_get_elev(tv, lat, lon, e) ? e : 0.0; // .unwrap_or(0.0), utils.rs:28-31,84
}

// find_normal, utils.rs:15-40.  The two calculators (azimuth 0 and 90) are handled by one rolled
// loop so the geodesic code is instantiated once, not four times (register pressure on the GPU).
ATMRT_HD Vec3 find_normal(const Earth& e, const TerrainView& tv, double lat, double lon) {
  const double DIFF = 15.0;
  double diff_ns = 0.0, diff_ew = 0.0;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll 1
#endif
  for (int k = 0; k < 2; k++) {
    DirCalc c;
    dircalc_new(e, lat, lon, k == 0 ? 0.0 : 90.0, c);
    double plat, plon, mlat, mlon;
    coords_at_dist(e, c, DIFF, plat, plon);   // p_north / p_east
    coords_at_dist(e, c, -DIFF, mlat, mlon);  // p_south / p_west
    double d = terrain_elev_or_zero(tv, plat, plon) - terrain_elev_or_zero(tv, mlat, mlon);
    if (k == 0) diff_ns = d;
    else diff_ew = d;
  }
  Vec3 dn, de, du;
  world_directions(e, lat, lon, dn, de, du);
  Vec3 vec_ns = dn * (2.0 * DIFF) + du * diff_ns;
  Vec3 vec_ew = de * (2.0 * DIFF) + du * diff_ew;
  Vec3 normal = cross(vec_ew, vec_ns);
  double len = dm_sqrt(normal.x * normal.x + normal.y * normal.y + normal.z * normal.z);
  return normal / len;
}

// ---------------------------------------------------------------------------------------------
// Ray stepper (crate atm-refraction 0.6, source absent): RK4 on the Snell / Bouguer ODE
// ---------------------------------------------------------------------------------------------

struct RayState { // RayState {x, h, dh}
  double x, h, dh;
};

struct Stepper {
  double x, a, b;  // flat: a = h, b = dh/dx; spherical: a = r, b = dr/dphi
  double h0, ang;  // straight rays: closed form from the start point
  int hint;        // atmosphere layer of the last n(h) evaluation (speed only, never changes a result)
};

ATMRT_HD void stepper_init(Stepper& s, bool spherical, double radius, double h0, double ang_rad) {
  s.x = 0.0;
  s.h0 = h0;
  s.ang = ang_rad;
  s.hint = 0;
  if (spherical) {
    s.a = h0 + radius;
    s.b = s.a * dm_tan(ang_rad);
  } else {
    s.a = h0;
    s.b = dm_tan(ang_rad);
  }
}

// The right-hand side proper.  FAST: n came from a certified interval (1 <= n <= 2^9, |dn| <= 2^15, and in the spherical model
// a = radius + h >= 1000 m, <= 2^25 m, both part of the certificate) and |b| <= 2^100, which keeps every operand and quotient
// of the three divisions inside dm_div's range; anything else divides in IEEE.
constexpr double ACCEL_FAST_MAX_B = 1.2676506002282294e30; // 2^100
// oracle/stepper.c accel_sph / accel_flat: the spherical form over one denominator, r'' = r + (2 r'^2 n + (r^2 + r'^2) r n') / (r n).
// FAST operand ranges: numerator terms <= 2^201 2^9 and 2^201 2^25 2^15, denominator in [1000, 2^34].
template <bool FAST>
ATMRT_HD double accel_rhs(bool spherical, double a, double b, double n, double dn) {
  if (spherical) {
    const double b2 = b * b, s = DM_FMA(a, a, b2);
    return a + div_sel<FAST>(DM_FMA(b2 + b2, n, s * a * dn), a * n);
  }
  return div_sel<FAST>(DM_FMA(b, b, 1.0) * dn, n);
}
// the flat form on a TIGHT segment: q = n - 1 <= 2^-10.5, so 1 - q is a seed of 1 / n with error q^2 <= 2^-21 (dm_div_seeded)
ATMRT_HD double accel_rhs_tight(bool spherical, double a, double b, double n, double q, double dn) {
  if (spherical) return accel_rhs<true>(true, a, b, n, dn);
  return dm_div_seeded(DM_FMA(b, b, 1.0) * dn, n, 1.0 - q);
}
// The generic right-hand side: per-lane layer search, IEEE operations — any atmosphere, any state.  On the GPU it runs for the few
// steps in which a wavefront straddles a layer boundary, and all the time only in pathological atmospheres.  (Inline: as a call it
// costs the hot path 2 % — live ranges split around the call site.)
template <bool CUBIC>
ATMRT_HD double ray_accel_generic(const AtmTable& a, bool spherical, double radius, double pa, double pb, int& hint) {
  const double eps = 0.01;
  const double h = spherical ? pa - radius : pa, h1 = h - eps, h2 = h + eps;
  const int k = atm_layer(a, h), k1 = atm_layer(a, h1), k2 = atm_layer(a, h2);
  hint = k; // the hint follows the lane, so that the fast path resumes once the wavefront is back inside one certified interval
  const double n = refr_n_layer<CUBIC, false>(a.k_refr, a.seg(k), h);
  const double n1 = refr_n_layer<CUBIC, false>(a.k_refr, a.seg(k1), h1);
  const double n2 = refr_n_layer<CUBIC, false>(a.k_refr, a.seg(k2), h2);
  const double dn = (n2 - n1) * REFR_INV_2EPS;
  return accel_rhs<false>(spherical, pa, pb, n, dn);
}

// The stepper's right-hand side.  When every active lane's three evaluation points (h, h -+ eps: 1 cm apart) lie in the certified
// part of the hinted layer and |b| <= 2^100 — the normal case: rays stay below 11 km in a physical atmosphere — one vote, one set of
// scalar layer parameters and the shortcut divisions serve the whole stage, and the instruction streams of the three evaluations
// interleave (the table look-ups of exp/log are the long latencies here).  Otherwise the generic version, same value.
// `fast` (wave-uniform) reports which version ran.
template <bool CUBIC>
ATMRT_HD double ray_accel(const AtmTable& atm, bool spherical, double radius, double a, double b, int& hint, bool& fast) {
  fast = false;
#if defined(__HIP_DEVICE_COMPILE__)
  const double eps = 0.01;
  const double h = spherical ? a - radius : a, h1 = h - eps, h2 = h + eps;
  const int ku = __builtin_amdgcn_readfirstlane(hint);
  // the table is read-only for the whole launch: reading it through the constant address space makes these scalar loads
  const AtmConstSeg ks = atm_const_seg(atm, ku);
  // the tight part of the hinted layer first (one vote: the normal case), its whole certified part when a lane is outside that
  const bool slope_ok = !(dm_fabs(b) > ACCEL_FAST_MAX_B);
  const bool tight = __all(h1 >= ks->tight_lo && h2 < ks->tight_hi && slope_ok);
  if (tight || __all(h1 >= ks->safe_lo && h2 < ks->safe_hi && slope_ok)) {
    const double k_refr = ks->k_refr, hb = ks->hb, tb = ks->tb, gtb = ks->gtb, ptb = ks->ptb, expo1 = ks->expo1, pb = ks->pb, lapse = ks->lapse, c2 = ks->c2, c3 = ks->c3, expo = ks->expo;
    const int cubic = ks->cubic, flags = tight ? ks->flags : ks->flags & ~ATM_SEG_TIGHT;
    const double exp_thr = ks->exp_thr;
    double n, n1, n2, q0;
    refr_n_layer3<CUBIC>(k_refr, cubic, flags, exp_thr, hb, tb, gtb, ptb, expo1, pb, lapse, c2, c3, expo, h, h1, h2, n, n1, n2, q0);
    const double dn = (n2 - n1) * REFR_INV_2EPS;
    fast = true;
    if ((flags & ATM_SEG_TIGHT) && !(CUBIC && cubic)) return accel_rhs_tight(spherical, a, b, n, q0, dn);
    return accel_rhs<true>(spherical, a, b, n, dn);
  }
#endif
  return ray_accel_generic<CUBIC>(atm, spherical, radius, a, b, hint);
}

// PathStepper::next: the state after one more step of `step` metres in x.  `accel(spherical, radius, a, b, hint, fast)` is
// the right-hand side of the ODE.  tame (wave-uniform on the GPU): all four stages ran certified (ray_accel's `fast`) — the start
// altitude then lies in a certified interval (radius + h >= 1000 m) and every stage slope is at most 2^100, so that the altitude
// change of the step is below 2^100 step: enough for calc_dist's shortcuts without another look at the end state.  Straight rays
// (closed form, no stages) report false; their callers test the altitudes (calc_dist_in_band).
// y + h/6 (k1 + 2 k2 + 2 k3 + k4) as oracle/stepper.c forms it
ATMRT_HD double rk4_sum(double y, double sixth, double k1, double k2, double k3, double k4) {
  return DM_FMA(sixth, DM_FMA(2.0, k3, DM_FMA(2.0, k2, k1)) + k4, y);
}
template <class Accel>
ATMRT_HD RayState stepper_next_with(Stepper& s, bool spherical, double radius, bool straight, double step, const Accel& accel, bool& tame) {
  RayState out;
  tame = false;
  if (straight) {
    s.x = s.x + step;
    if (spherical) {
      double r0 = s.h0 + radius;
      double phi = s.x / radius;
      double c = dm_cos(s.ang + phi);
      out.h = r0 * dm_cos(s.ang) / c - radius;
      out.dh = dm_tan(s.ang + phi);
    } else {
      out.h = s.h0 + s.x * dm_tan(s.ang);
      out.dh = dm_tan(s.ang);
    }
    out.x = s.x;
    return out;
  }
  double d = spherical ? step / radius : step;
  double half = 0.5 * d, sixth = d / 6.0;
  double a = s.a, b = s.b;
  bool f1, f2, f3, f4;
  double k1a = b;
  double k1b = accel(spherical, radius, a, b, s.hint, f1);
  double k2a = DM_FMA(half, k1b, b);
  double k2b = accel(spherical, radius, DM_FMA(half, k1a, a), k2a, s.hint, f2);
  double k3a = DM_FMA(half, k2b, b);
  double k3b = accel(spherical, radius, DM_FMA(half, k2a, a), k3a, s.hint, f3);
  double k4a = DM_FMA(d, k3b, b);
  double k4b = accel(spherical, radius, DM_FMA(d, k3a, a), k4a, s.hint, f4);
  tame = f1 && f2 && f3 && f4;
  s.a = rk4_sum(a, sixth, k1a, k2a, k3a, k4a);
  s.b = rk4_sum(b, sixth, k1b, k2b, k3b, k4b);
  s.x = s.x + step;
  out.x = s.x;
  if (spherical) {
    out.h = s.a - radius;
    out.dh = s.b / radius;
  } else {
    out.h = s.a;
    out.dh = s.b;
  }
  return out;
}

template <bool CUBIC>
struct SerialAccel {
  const AtmTable& atm;
  ATMRT_HD double operator()(bool spherical, double radius, double a, double b, int& hint, bool& fast) const {
    return ray_accel<CUBIC>(atm, spherical, radius, a, b, hint, fast);
  }
};
template <bool CUBIC = true>
ATMRT_HD RayState stepper_next(Stepper& s, const AtmTable& atm, bool spherical, double radius, bool straight, double step, bool& tame) {
  return stepper_next_with(s, spherical, radius, straight, step, SerialAccel<CUBIC>{atm}, tame);
}
template <bool CUBIC = true>
ATMRT_HD RayState stepper_next(Stepper& s, const AtmTable& atm, bool spherical, double radius, bool straight, double step) {
  bool tame;
  return stepper_next_with(s, spherical, radius, straight, step, SerialAccel<CUBIC>{atm}, tame);
}
ATMRT_HD bool atm_has_cubic(const AtmTable& a) {
  for (int k = 0; k < a.n; k++)
    if (a.seg(k).cubic) return true;
  return false;
}

// calc_dist, utils.rs:42-53
// `fast` (wave-uniform on the GPU): the step was tame (stepper_next_with: h0 + radius >= 1000 m, |dh| <= 2^100 step), or — straight
// rays — both altitudes lie in [AtmTable::alt_lo, alt_hi].  Every caller advances by one simulation step (dx > 0, 1 mm .. 1e8 m or
// nothing is certified; likewise the radius), so the radicand then is inside dm_sqrt_inrange's range: at most 2^260, and at least
// dx^2 (flat), or (spherical) dh^2 >= 1e6 when the mean radius fell below half the start radius, (dx / radius * 500)^2 otherwise.
// Anything else (a ray that left for 1e300 m, NaN): the IEEE square root.
ATMRT_HD bool calc_dist_in_band(const AtmTable& atm, double h) {
#if defined(__HIP_DEVICE_COMPILE__)
  const AtmConstTable ka = atm_const_table(atm);
  return h >= ka->alt_lo && h <= ka->alt_hi;
#else
  return h >= atm.alt_lo && h <= atm.alt_hi;
#endif
}
// rradius: RN(1 / radius), tabulated by the host (Frame::inv_shape_radius), or 0: the quotient dx / radius then comes from dm_div_r's
// two corrections of dx * rradius instead of dm_div's v_rcp_f64 and Newton steps — the same IEEE quotient, six issue slots fewer.
ATMRT_HD double calc_dist(bool spherical, double radius, double x0, double h0, double x1, double h1, bool fast, double rradius = 0.0) {
  double dx = x1 - x0;
  double dh = h1 - h0;
  if (fast) {
    if (!spherical) return dm_sqrt_inrange(dx * dx + dh * dh);
    double avg_h = (h1 + h0) / 2.0;
    double dx2 = (rradius != 0.0 ? dm_div_r(dx, radius, rradius) : dm_div(dx, radius)) * (avg_h + radius);
    return dm_sqrt_inrange(dx2 * dx2 + dh * dh);
  }
  if (!spherical) return dm_sqrt(dx * dx + dh * dh);
  double avg_h = (h1 + h0) / 2.0;
  double dx2 = dx / radius * (avg_h + radius);
  return dm_sqrt(dx2 * dx2 + dh * dh);
}

// ---------------------------------------------------------------------------------------------
// Pixel -> ray mapping
// ---------------------------------------------------------------------------------------------

// get_ray_elev / get_ray_dir, fast.rs:111-125 (i16 casts; `as` binds tighter than `/`)
ATMRT_HD double fast_ray_elev(const atmrt_params_t& p, int y) {
  double width = (double)p.width, height = (double)p.height;
  double aspect = width / height;
  double yy = (double)(int16_t)((int16_t)y - (int16_t)p.height / 2) / height;
  return p.frame.tilt - yy * p.frame.fov / aspect;
}
ATMRT_HD double fast_ray_dir(const atmrt_params_t& p, int x) {
  double width = (double)p.width;
  double xx = (double)(int16_t)((int16_t)x - (int16_t)p.width / 2) / width;
  return p.frame.direction + xx * p.frame.fov;
}

// Pinhole camera set-up shared by every pixel of a frame: rows of Rz(yaw) Ry(pitch) Rx(roll) as
// nalgebra's Rotation3::from_euler_angles(0, -tilt, direction) builds them, and the focal length.
struct Pinhole {
  double m[9];
  double z;
};
ATMRT_HD void pinhole_init(const atmrt_params_t& p, Pinhole& ph) {
  double roll = 0.0, pitch = -dm_to_radians(p.frame.tilt), yaw = dm_to_radians(p.frame.direction);
  double sr, cr, sp, cp, sy, cy;
  dm_sincos(roll, &sr, &cr);
  dm_sincos(pitch, &sp, &cp);
  dm_sincos(yaw, &sy, &cy);
  ph.m[0] = cy * cp; ph.m[1] = cy * sp * sr - sy * cr; ph.m[2] = cy * sp * cr + sy * sr;
  ph.m[3] = sy * cp; ph.m[4] = sy * sp * sr + cy * cr; ph.m[5] = sy * sp * cr - cy * sr;
  ph.m[6] = -sp;     ph.m[7] = cp * sr;                ph.m[8] = cp * cr;
  ph.z = (double)p.width / 2.0 / dm_tan(dm_to_radians(p.frame.fov) / 2.0);
}
// RectilinearGenerator::get_ray_params, rectilinear.rs:78-100: direction and elevation in RADIANS
ATMRT_HD void rect_ray_params(const atmrt_params_t& p, const Pinhole& ph, int px, int py, double& direction,
                              double& elevation) {
  double x = (double)(int16_t)((int16_t)px - (int16_t)p.width / 2);
  double y = (double)(int16_t)((int16_t)py - (int16_t)p.height / 2);
  double vx = ph.z, vy = x, vz = -y; // Vector3::new(z, x, -y)
  double dx = ph.m[0] * vx + ph.m[1] * vy + ph.m[2] * vz;
  double dy = ph.m[3] * vx + ph.m[4] * vy + ph.m[5] * vz;
  double dz = ph.m[6] * vx + ph.m[7] * vy + ph.m[8] * vz;
  double len = dm_sqrt(dx * dx + dy * dy + dz * dz);
  dx /= len;
  dy /= len;
  dz /= len;
  direction = dm_atan2(dy, dx);
  elevation = dm_asin(dz);
}

// TracingState::interpolate for one scalar, utils.rs:108-125
ATMRT_HD double lerp_ts(double a, double b, double prop) { return a + (b - a) * prop; }

} // namespace atmrt

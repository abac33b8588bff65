/* atmosphere.c — layered atmosphere + refractive index of air.  ORACLE (test infrastructure).
 *
 * Restates what the reference obtains from crate `atm-refraction` 0.6 (source absent, PARITY
 * UNPINNED): `Atmosphere::from_def` (params.rs:514), `atmosphere.temperature/pressure`
 * (atm_printer.rs:37-46), `Environment::n` (renderer/mod.rs:425) and the dn/dh the ray ODE needs.
 * Published models used (choices recorded in DESIGN.md):
 *   - temperature: the functions of the YAML schema (README.md:283-323): `Linear{gradient}` and `Spline` (interpolating
 *     cubic spline through the points with Natural / Derivatives / SecondDerivatives end conditions, continued linearly
 *     with its end slope outside its knots); Linear functions take their level from the temperature fixed point or from
 *     continuity with a neighbouring function; `AtmosphereDef::us_76` = U.S. Standard Atmosphere 1976 layers 0-86 km;
 *   - pressure: hydrostatic equilibrium of an ideal gas, p = p_b exp(-(g0 M/R) Int dh/T): closed form on Linear
 *     functions (NOAA-S/T 76-1562 eq. 33a/33b), 5-point Gauss-Legendre quadrature per spline interval;
 *     g0 = 9.80665 m/s2, M = 0.0289644 kg/mol, R* = 8.31432 J/(mol K);
 *   - refractive index: Ciddor (Appl. Opt. 35, 1566, 1996) as documented by NIST's Engineering
 *     Metrology Toolbox, dry air (the YAML schema has no humidity), x_CO2 = 450 umol/mol;
 *   - dn/dh: central difference with eps = 0.01 m, (n(h + eps) - n(h - eps)) * (0.5 / eps)  (0.5 / 0.01 is 50.0 exactly).
 * EVALUATION ORDER (round 4).  The crate's expression order is unknown, so the roundings inside these formulas are this build's to
 * fix; they are fixed HERE, and the product follows bit for bit.  Where the published formula has the shape a*b + c it is one fused
 * multiply-add (C99 fma, exactly rounded in either flavour):
 *     dh = h - hb;  x = T / tb = fma(lapse / tb, dh, 1);  T = tb x  (the barometric formula's own variable, p = pb x^expo);
 *     n(h) on a Linear segment takes the density term directly, pt = p / T = (pb / tb) x^(expo - 1), and t = fma(tb, x, -273.15);
 *     Z = fma(pt, fma(pt, d, -A), 1),  A = fma(t, fma(t, a2, a1), a0);   n = 1 + k pt / Z.
 * Against the round-3 order (every operation rounded separately, T / tb by division) the outputs of a frame move by rounding
 * errors only: tests/test_oracle_known_answers.py compares both with the modular forms (n from pressure(h) / temperature(h); the two-term right-hand
 * side and textbook RK4 sums) in multi-precision arithmetic, and the golden vectors written under the round-3 order are still
 * reproduced to 1e-9 (tests/test_golden.py: deliberately not regenerated).
 */
#include "oracle.h"
#include "oracle_math.h"

#include <stdlib.h>
#include <string.h>

#define G0 9.80665
#define M_AIR 0.0289644
#define R_GAS 8.31432

void oracle_atmosphere_us76(atmrt_atmosphere_t* a) {
  static const double alt[7] = {0.0, 11000.0, 20000.0, 32000.0, 47000.0, 51000.0, 71000.0};
  static const double lapse[7] = {-0.0065, 0.0, 0.001, 0.0028, 0.0, -0.0028, -0.002};
  static atmrt_temp_function_t fns[7];
  int k;
  memset(a, 0, sizeof *a);
  a->pressure_altitude = 0.0;
  a->pressure = 101325.0;
  a->temperature_altitude = 0.0;
  a->temperature = 288.15;
  a->has_temperature_fixed_point = 1;
  a->n_functions = 7;
  for (k = 0; k < 7; k++) { /* the same values on every call: safe to rewrite */
    fns[k].kind = ATMRT_TEMP_LINEAR;
    fns[k].altitude = alt[k];
    fns[k].gradient = lapse[k];
  }
  a->functions = fns;
}

static int layer_of(const oracle_env_atm* a, double h) {
  int k;
  for (k = a->n - 1; k >= 1; k--)
    if (h >= a->from[k]) return k;
  return 0;
}

static double cubic_temperature(const oracle_env_atm* a, int k, double dh) {
  return a->tb[k] + dh * (a->lapse[k] + dh * (a->c2[k] + dh * a->c3[k]));
}

static double seg_temperature(const oracle_env_atm* a, int k, double h) {
  if (a->cubic[k]) return cubic_temperature(a, k, h - a->hb[k]);
  return a->tb[k] * om_fma(a->lapse[k] / a->tb[k], h - a->hb[k], 1.0);
}

/* Int_{hb}^{hb+dh} dh'/T(h'): 5-point Gauss-Legendre */
static double inv_t_integral(const oracle_env_atm* a, int k, double dh) {
  const double x1 = 0.5384693101056831, x2 = 0.9061798459386640;
  const double w0 = 0.5688888888888889, w1 = 0.4786286704993665, w2 = 0.2369268850561891;
  double half = 0.5 * dh;
  double s = w0 / cubic_temperature(a, k, half);
  s += w1 / cubic_temperature(a, k, half - half * x1);
  s += w1 / cubic_temperature(a, k, half + half * x1);
  s += w2 / cubic_temperature(a, k, half - half * x2);
  s += w2 / cubic_temperature(a, k, half + half * x2);
  return half * s;
}

/* p(h)/pb of segment k */
static double pressure_ratio(const oracle_env_atm* a, int k, double h) {
  if (a->cubic[k]) return om_exp(a->expo[k] * inv_t_integral(a, k, h - a->hb[k]));
  if (a->lapse[k] != 0.0) { /* T / tb = 1 + (lapse / tb) (h - hb) */
    double x = om_fma(a->lapse[k] / a->tb[k], h - a->hb[k], 1.0);
    return om_pow(x, a->expo[k]);
  }
  return om_exp(a->expo[k] * (h - a->hb[k]));
}

/* interpolating cubic spline: second derivatives m[] by the Thomas algorithm */
static void spline_second_derivatives(const atmrt_temp_function_t* fn, double* m) {
  const int np = fn->n_points;
  const double* x = fn->point_altitude;
  const double* y = fn->point_temperature;
  double* cp = (double*)malloc(2 * (size_t)np * sizeof(double));
  double* dp = cp + np;
  double b0, c0, d0, an, bn, dn;
  int i;
  if (fn->boundary == ATMRT_SPLINE_DERIVATIVES) {
    double h0 = x[1] - x[0], hn = x[np - 1] - x[np - 2];
    b0 = 2.0 * h0; c0 = h0; d0 = 6.0 * ((y[1] - y[0]) / h0 - fn->bc[0]);
    an = hn; bn = 2.0 * hn; dn = 6.0 * (fn->bc[1] - (y[np - 1] - y[np - 2]) / hn);
  } else {
    b0 = 1.0; c0 = 0.0; d0 = fn->boundary == ATMRT_SPLINE_SECOND_DERIVATIVES ? fn->bc[0] : 0.0;
    an = 0.0; bn = 1.0; dn = fn->boundary == ATMRT_SPLINE_SECOND_DERIVATIVES ? fn->bc[1] : 0.0;
  }
  cp[0] = c0 / b0;
  dp[0] = d0 / b0;
  for (i = 1; i < np; i++) {
    double ai, bi, ci, di, den;
    if (i < np - 1) {
      double hl = x[i] - x[i - 1], hr = x[i + 1] - x[i];
      ai = hl; bi = 2.0 * (hl + hr); ci = hr;
      di = 6.0 * ((y[i + 1] - y[i]) / hr - (y[i] - y[i - 1]) / hl);
    } else {
      ai = an; bi = bn; ci = 0.0; di = dn;
    }
    den = bi - ai * cp[i - 1];
    cp[i] = ci / den;
    dp[i] = (di - ai * dp[i - 1]) / den;
  }
  m[np - 1] = dp[np - 1];
  for (i = np - 2; i >= 0; i--) m[i] = dp[i] - cp[i] * m[i + 1];
  free(cp);
}

void oracle_atm_free(oracle_env_atm* a) {
  free(a->hb); /* one block: see oracle_atm_compile */
  free(a->cubic);
  memset(a, 0, sizeof *a);
}

int oracle_atm_compile(const atmrt_atmosphere_t* def, double wavelength, oracle_env_atm* out) {
  const double gmr = G0 * M_AIR / R_GAS;
  const int nf = def->n_functions;
  int n = 0, j, k, pass, any = 0, jp;
  int *first_seg, *anchored;
  size_t cap = 0;
  double* m = NULL;
  memset(out, 0, sizeof *out);
  if (nf < 1 || !def->functions) return -1;
  for (j = 0; j < nf; j++) { /* a Linear function is one segment, a Spline at most its knot intervals + two continuations */
    const atmrt_temp_function_t* fn = &def->functions[j];
    if (fn->kind == ATMRT_TEMP_SPLINE && (fn->n_points < 2 || !fn->point_altitude || !fn->point_temperature)) return -4;
    cap += fn->kind == ATMRT_TEMP_SPLINE ? (size_t)fn->n_points + 1 : 1;
  }
  out->hb = (double*)calloc(8 * cap, sizeof(double));
  out->tb = out->hb + cap, out->pb = out->tb + cap, out->lapse = out->pb + cap, out->from = out->lapse + cap;
  out->expo = out->from + cap, out->c2 = out->expo + cap, out->c3 = out->c2 + cap;
  out->cubic = (int*)calloc(cap + 2 * ((size_t)nf + 1), sizeof(int));
  first_seg = out->cubic + cap; /* scratch behind the segment flags, released with them */
  anchored = first_seg + nf + 1;
  for (j = 2; j < nf; j++)
    if (!(def->functions[j].altitude > def->functions[j - 1].altitude)) return -2;
  for (j = 0; j < nf; j++) {
    const atmrt_temp_function_t* fn = &def->functions[j];
    const int has_lo = j > 0, has_hi = j + 1 < nf;
    const double lo = has_lo ? fn->altitude : 0.0, hi = has_hi ? def->functions[j + 1].altitude : 0.0;
    first_seg[j] = n;
    anchored[j] = 0;
    if (fn->kind == ATMRT_TEMP_LINEAR) {
      out->from[n] = lo;
      out->lapse[n] = fn->gradient;
      n++;
      continue;
    }
    if (fn->kind != ATMRT_TEMP_SPLINE) {
      free(m);
      return -1;
    }
    {
      const int np = fn->n_points;
      const double* x = fn->point_altitude;
      const double* y = fn->point_temperature;
      int i;
      for (i = 1; i < np; i++)
        if (!(x[i] > x[i - 1])) {
          free(m);
          return -4;
        }
      free(m);
      m = (double*)malloc((size_t)np * sizeof(double));
      spline_second_derivatives(fn, m);
      if (!has_lo || lo < x[0]) { /* linear continuation below the first knot, slope S'(x0) */
        double hh = x[1] - x[0];
        out->from[n] = lo;
        out->hb[n] = x[0];
        out->tb[n] = y[0];
        out->lapse[n] = (y[1] - y[0]) / hh - hh * (2.0 * m[0] + m[1]) / 6.0;
        n++;
      }
      for (i = 0; i + 1 < np; i++) {
        double hh;
        if (has_hi && x[i] >= hi) break;
        if (has_lo && x[i + 1] <= lo) continue;
        hh = x[i + 1] - x[i];
        out->from[n] = (has_lo && lo > x[i]) ? lo : x[i];
        out->hb[n] = x[i];
        out->tb[n] = y[i];
        out->lapse[n] = (y[i + 1] - y[i]) / hh - hh * (2.0 * m[i] + m[i + 1]) / 6.0;
        out->c2[n] = m[i] / 2.0;
        out->c3[n] = (m[i + 1] - m[i]) / (6.0 * hh);
        out->cubic[n] = 1;
        n++;
      }
      if (!has_hi || hi > x[np - 1]) { /* linear continuation above the last knot, slope S'(x_last) */
        double hh = x[np - 1] - x[np - 2];
        out->from[n] = (has_lo && lo > x[np - 1]) ? lo : x[np - 1]; /* every knot below the function's own start: it still begins at `lo` */
        out->hb[n] = x[np - 1];
        out->tb[n] = y[np - 1];
        out->lapse[n] = (y[np - 1] - y[np - 2]) / hh + hh * (m[np - 2] + 2.0 * m[np - 1]) / 6.0;
        n++;
      }
      anchored[j] = 1;
    }
  }
  free(m);
  first_seg[nf] = n;
  out->n = n;
  out->from[0] = 0.0;
  /* absolute temperature of the Linear functions */
  if (def->has_temperature_fixed_point) {
    int jt = 0;
    for (j = nf - 1; j >= 1; j--)
      if (def->temperature_altitude >= def->functions[j].altitude) { jt = j; break; }
    if (def->functions[jt].kind == ATMRT_TEMP_LINEAR && !anchored[jt]) {
      k = first_seg[jt];
      out->hb[k] = jt == 0 ? def->temperature_altitude : out->from[k];
      out->tb[k] = def->temperature - out->lapse[k] * (def->temperature_altitude - out->hb[k]);
      anchored[jt] = 1;
    }
  }
  for (j = 0; j < nf; j++) any = any || anchored[j];
  if (!any) return -3;
  for (pass = 0; pass < nf; pass++) {
    for (j = 0; j < nf; j++) {
      if (anchored[j]) continue;
      k = first_seg[j];
      if (j > 0 && anchored[j - 1]) {
        int kl = first_seg[j] - 1;
        out->hb[k] = out->from[k];
        out->tb[k] = seg_temperature(out, kl, out->from[k]);
        anchored[j] = 1;
      } else if (j + 1 < nf && anchored[j + 1]) {
        int ku = first_seg[j + 1];
        double top = def->functions[j + 1].altitude;
        out->hb[k] = j == 0 ? top : out->from[k];
        out->tb[k] = seg_temperature(out, ku, top) - out->lapse[k] * (top - out->hb[k]);
        anchored[j] = 1;
      }
    }
  }
  for (k = 0; k < n; k++)
    out->expo[k] = out->cubic[k] ? -gmr : (out->lapse[k] != 0.0 ? -gmr / out->lapse[k] : -gmr / out->tb[k]);
  /* pressure: chain outwards from the pressure fixed point */
  jp = layer_of(out, def->pressure_altitude);
  out->pb[jp] = def->pressure / pressure_ratio(out, jp, def->pressure_altitude);
  for (k = jp + 1; k < n; k++) {
    double pk = out->pb[k - 1] * pressure_ratio(out, k - 1, out->from[k]);
    out->pb[k] = pk / pressure_ratio(out, k, out->from[k]);
  }
  for (k = jp - 1; k >= 0; k--) {
    double pk = out->pb[k + 1] * pressure_ratio(out, k + 1, out->from[k + 1]);
    out->pb[k] = pk / pressure_ratio(out, k, out->from[k + 1]);
  }
  {
    /* Ciddor 1996, dry air: (n - 1) = (rho_a / rho_axs) * r_axs,  rho = p*Ma/(Z*R*T) */
    const double k0 = 238.0185, k1 = 5792105.0, k2 = 57.362, k3 = 167917.0;
    const double xco2 = 450.0, pr1 = 101325.0, tr1 = 288.15, za = 0.9995922115, r = 8.314472;
    double lam_um = wavelength * 1.0e6;
    double s = 1.0 / (lam_um * lam_um);
    double ras = 1.0e-8 * (k1 / (k0 - s) + k3 / (k2 - s));
    double raxs = ras * (1.0 + 5.34e-7 * (xco2 - 450.0));
    double ma = 0.0289635 + 1.2011e-8 * (xco2 - 400.0);
    double rho_axs = pr1 * ma / (za * r * tr1);
    out->k_refr = raxs / rho_axs * ma / r;
  }
  return 0;
}

double oracle_atm_temperature(const oracle_env_atm* a, double h) { return seg_temperature(a, layer_of(a, h), h); }

double oracle_atm_pressure(const oracle_env_atm* a, double h) {
  int k = layer_of(a, h);
  return a->pb[k] * pressure_ratio(a, k, h);
}

/* Environment::n(h) */
double oracle_n(const oracle_env_atm* a, double h) {
  int k = layer_of(a, h);
  double pt, t, aa, z;
  const double a0 = 1.58123e-6, a1 = -2.9331e-8, a2 = 1.1043e-10, d = 1.83e-11;
  if (a->cubic[k]) {
    double temp = seg_temperature(a, k, h);
    pt = a->pb[k] * pressure_ratio(a, k, h) / temp;
    t = temp - 273.15;
  } else {
    /* Linear segment: Ciddor needs the air density, i.e. p / T, and on a Linear segment that is itself a closed form,
     * p / T = (pb / tb) x^(expo - 1) with x = T / tb  (isothermal: (pb / tb) exp(expo dh)) — no division per evaluation */
    double dh = h - a->hb[k];
    double x = om_fma(a->lapse[k] / a->tb[k], dh, 1.0);
    double ptb = a->pb[k] / a->tb[k];
    pt = a->lapse[k] != 0.0 ? ptb * om_pow(x, a->expo[k] - 1.0) : ptb * om_exp(a->expo[k] * dh);
    t = om_fma(a->tb[k], x, -273.15);
  }
  aa = om_fma(t, om_fma(t, a2, a1), a0);
  z = om_fma(pt, om_fma(pt, d, -aa), 1.0); /* 1 - pt (a0 + a1 t + a2 t^2) + pt^2 d */
  return 1.0 + a->k_refr * pt / z;
}

double oracle_dn(const oracle_env_atm* a, double h) {
  const double eps = 0.01;
  double n1 = oracle_n(a, h - eps);
  double n2 = oracle_n(a, h + eps);
  return (n2 - n1) * (0.5 / eps);
}

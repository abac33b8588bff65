/* atmosphere.c — layered atmosphere + refractive index of air.  ORACLE (test infrastructure).
 *
 * Restates what the reference obtains from crate `atm-refraction` 0.6 (source absent, PARITY
 * UNPINNED): `Atmosphere::from_def` (params.rs:514), `atmosphere.temperature/pressure`
 * (atm_printer.rs:37-46), `Environment::n` (renderer/mod.rs:425) and the dn/dh the ray ODE needs.
 * Published models used (choices recorded in DESIGN.md):
 *   - temperature: piecewise-linear lapse-rate layers, schema of README.md:283-323;
 *     `AtmosphereDef::us_76` = U.S. Standard Atmosphere 1976 layers 0-86 km (NOAA-S/T 76-1562);
 *   - pressure: hydrostatic equilibrium of an ideal gas, closed form per layer (same document,
 *     eq. 33a/33b) with g0 = 9.80665 m/s2, M = 0.0289644 kg/mol, R* = 8.31432 J/(mol K);
 *   - refractive index: Ciddor (Appl. Opt. 35, 1566, 1996) as documented by NIST's Engineering
 *     Metrology Toolbox, dry air (the YAML schema has no humidity), x_CO2 = 450 umol/mol;
 *   - dn/dh: central difference with eps = 0.01 m.
 */
#include "oracle.h"
#include "oracle_math.h"

#define G0 9.80665
#define M_AIR 0.0289644
#define R_GAS 8.31432

void oracle_atmosphere_us76(atmrt_atmosphere_t* a) {
  static const double alt[7] = {0.0, 11000.0, 20000.0, 32000.0, 47000.0, 51000.0, 71000.0};
  static const double lapse[7] = {-0.0065, 0.0, 0.001, 0.0028, 0.0, -0.0028, -0.002};
  int k;
  a->pressure_altitude = 0.0;
  a->pressure = 101325.0;
  a->temperature_altitude = 0.0;
  a->temperature = 288.15;
  a->n_layers = 7;
  a->_pad = 0;
  for (k = 0; k < ATMRT_MAX_ATM_LAYERS; k++) {
    a->layer_altitude[k] = k < 7 ? alt[k] : 0.0;
    a->layer_gradient[k] = k < 7 ? lapse[k] : 0.0;
  }
}

static int layer_of(const oracle_env_atm* a, double h) {
  int k;
  for (k = a->n - 1; k >= 1; k--)
    if (h >= a->from[k]) return k;
  return 0;
}

/* p(h)/pb of layer k */
static double pressure_ratio(const oracle_env_atm* a, int k, double h) {
  if (a->lapse[k] != 0.0) {
    double t = a->tb[k] + a->lapse[k] * (h - a->hb[k]);
    return om_pow(t / a->tb[k], a->expo[k]);
  }
  return om_exp(a->expo[k] * (h - a->hb[k]));
}

int oracle_atm_compile(const atmrt_atmosphere_t* def, double wavelength, oracle_env_atm* out) {
  const double gmr = G0 * M_AIR / R_GAS;
  int n = def->n_layers, k, jt, jp;
  if (n < 1 || n > ATMRT_MAX_ATM_LAYERS) return -1;
  out->n = n;
  for (k = 0; k < n; k++) {
    out->lapse[k] = def->layer_gradient[k];
    out->from[k] = k == 0 ? 0.0 : def->layer_altitude[k];
    if (k >= 2 && !(out->from[k] > out->from[k - 1])) return -1;
  }
  /* temperature: chain outwards from the layer that holds the fixed point */
  jt = 0;
  for (k = n - 1; k >= 1; k--)
    if (def->temperature_altitude >= out->from[k]) { jt = k; break; }
  out->hb[jt] = jt == 0 ? def->temperature_altitude : out->from[jt];
  out->tb[jt] = def->temperature - out->lapse[jt] * (def->temperature_altitude - out->hb[jt]);
  for (k = jt + 1; k < n; k++) {
    out->hb[k] = out->from[k];
    out->tb[k] = out->tb[k - 1] + out->lapse[k - 1] * (out->from[k] - out->hb[k - 1]);
  }
  for (k = jt - 1; k >= 0; k--) {
    /* layer k ends at from[k+1], where layer k+1 has temperature tb[k+1] */
    out->hb[k] = k == 0 ? out->from[1] : out->from[k];
    out->tb[k] = out->tb[k + 1] - out->lapse[k] * (out->from[k + 1] - out->hb[k]);
  }
  for (k = 0; k < n; k++)
    out->expo[k] = out->lapse[k] != 0.0 ? -gmr / out->lapse[k] : -gmr / out->tb[k];
  /* pressure: same chaining from the pressure fixed point */
  jp = 0;
  for (k = n - 1; k >= 1; k--)
    if (def->pressure_altitude >= out->from[k]) { jp = k; break; }
  out->pb[jp] = def->pressure / pressure_ratio(out, jp, def->pressure_altitude);
  for (k = jp + 1; k < n; k++) out->pb[k] = out->pb[k - 1] * pressure_ratio(out, k - 1, out->from[k]);
  for (k = jp - 1; k >= 0; k--) out->pb[k] = out->pb[k + 1] / pressure_ratio(out, k, out->from[k + 1]);
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

double oracle_atm_temperature(const oracle_env_atm* a, double h) {
  int k = layer_of(a, h);
  return a->tb[k] + a->lapse[k] * (h - a->hb[k]);
}

double oracle_atm_pressure(const oracle_env_atm* a, double h) {
  int k = layer_of(a, h);
  return a->pb[k] * pressure_ratio(a, k, h);
}

/* Environment::n(h) */
double oracle_n(const oracle_env_atm* a, double h) {
  const double a0 = 1.58123e-6, a1 = -2.9331e-8, a2 = 1.1043e-10, d = 1.83e-11;
  int k = layer_of(a, h);
  double temp = a->tb[k] + a->lapse[k] * (h - a->hb[k]);
  double p = a->pb[k] * pressure_ratio(a, k, h);
  double t = temp - 273.15;
  double pt = p / temp;
  double z = 1.0 - pt * (a0 + t * (a1 + t * a2)) + pt * pt * d;
  return 1.0 + a->k_refr * pt / z;
}

double oracle_dn(const oracle_env_atm* a, double h) {
  const double eps = 0.01;
  double n1 = oracle_n(a, h - eps);
  double n2 = oracle_n(a, h + eps);
  return (n2 - n1) / (2.0 * eps);
}

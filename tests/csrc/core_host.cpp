// Test-only host build of the PRODUCT's numerics header (atm-raytracer_amd/csrc/atmrt_core.h) so that the
// operation order of the HIP kernels' device functions can be compared bit-for-bit with the oracle on the CPU,
// without a GPU.  Never linked into the product; the product has no CPU path.
#include "../../atm-raytracer_amd/csrc/atmrt_core.h"
#include <vector>
using namespace atmrt;

extern "C" {
int ch_atm(const atmrt_atmosphere_t* def, double wavelength, size_t n, const double* h, double* t, double* p, double* nn, double* dn) {
  AtmTableBuf buf;
  if (atm_compile(*def, wavelength, buf)) return -1;
  const AtmTable& a = buf.table();
  for (size_t i = 0; i < n; i++) { t[i] = atm_temperature(a, h[i]); p[i] = atm_pressure(a, h[i]); nn[i] = refr_n(a, h[i]); dn[i] = refr_dn(a, h[i]); }
  return 0;
}
int ch_coords(const atmrt_earth_model_t* m, double lat0, double lon0, double dir, size_t n, const double* d, double* lat, double* lon) {
  Earth e;
  if (earth_resolve(*m, e)) return -1;
  DirCalc c;
  dircalc_new(e, lat0, lon0, dir, c);
  for (size_t i = 0; i < n; i++) coords_at_dist(e, c, d[i], lat[i], lon[i]);
  return 0;
}
int ch_cart(const atmrt_earth_model_t* m, double lat, double lon, double elev, double* out12) {
  Earth e;
  if (earth_resolve(*m, e)) return -1;
  Vec3 c = as_cartesian(e, lat, lon, elev), n, ea, up;
  world_directions(e, lat, lon, n, ea, up);
  double v[12] = {c.x, c.y, c.z, n.x, n.y, n.z, ea.x, ea.y, ea.z, up.x, up.y, up.z};
  for (int i = 0; i < 12; i++) out12[i] = v[i];
  return 0;
}
// atm_certify: the altitude interval of every segment inside which the GPU may take its shortcut divisions, and an independent check
// of the certificate: min T, max p/T, max |Z - 1| and max n over a dense sample of the interval
int ch_certify(const atmrt_atmosphere_t* def, double wavelength, int spherical, double radius, double step, int* n_seg, double* from,
               double* safe_lo, double* safe_hi, double* band2, double* min_t, double* max_pt, double* max_z_dev, double* max_n,
               double* max_abs_e, double* flags, double* max_dt_rel, double* max_dz_rel, double* tight_lo, double* tight_hi, double* tight_max_q, double* tight_max_zdev) {
  AtmTableBuf buf;
  if (atm_compile(*def, wavelength, buf)) return -1;
  AtmTable& a = buf.table();
  atm_certify(a, spherical != 0, radius, step);
  *n_seg = a.n;
  band2[0] = a.alt_lo;
  band2[1] = a.alt_hi;
  for (int k = 0; k < a.n; k++) {
    from[k] = a.seg(k).from;
    safe_lo[k] = a.seg(k).safe_lo;
    safe_hi[k] = a.seg(k).safe_hi;
    min_t[k] = 1e300; max_pt[k] = max_z_dev[k] = max_n[k] = max_abs_e[k] = max_dt_rel[k] = max_dz_rel[k] = 0.0;
    flags[k] = (double)a.seg(k).flags;
    tight_lo[k] = a.seg(k).tight_lo;
    tight_hi[k] = a.seg(k).tight_hi;
    tight_max_q[k] = tight_max_zdev[k] = 0.0;
    if (!(a.seg(k).safe_lo < a.seg(k).safe_hi)) continue;
    const int N = 4000;
    for (int i = 0; i <= N; i++) {
      double h = a.seg(k).safe_lo + (a.seg(k).safe_hi - a.seg(k).safe_lo) * i / N;
      double t = atm_seg_temperature(a, k, h), pr = a.seg(k).pb * atm_pressure_ratio(a, k, h), pt = pr / t, c = t - 273.15;
      double z = 1.0 - pt * (1.58123e-6 + c * (-2.9331e-8 + c * 1.1043e-10)) + pt * pt * 1.83e-11;
      double n = refr_n_layer(a.k_refr, a.seg(k), h);
      if (!(t >= min_t[k])) min_t[k] = t;
      if (!(pt <= max_pt[k])) max_pt[k] = pt;
      double zd = z > 1.0 ? z - 1.0 : 1.0 - z;
      if (!(zd <= max_z_dev[k])) max_z_dev[k] = zd;
      if (!(n <= max_n[k])) max_n[k] = n;
      if (h >= a.seg(k).tight_lo && h < a.seg(k).tight_hi) {
        if (!(n - 1.0 <= tight_max_q[k])) tight_max_q[k] = n - 1.0;
        if (!(zd <= tight_max_zdev[k])) tight_max_zdev[k] = zd;
      }
      // what a TIGHT segment promises about the three points of one right-hand side, h and h -+ 1 cm (atm_interval_tight)
      for (int sgn = -1; sgn <= 1 && h >= a.seg(k).tight_lo && h < a.seg(k).tight_hi; sgn += 2) {
        const double h2 = h + sgn * 0.01, t2 = atm_seg_temperature(a, k, h2), pt2 = a.seg(k).pb * atm_pressure_ratio(a, k, h2) / t2, c2 = t2 - 273.15;
        const double z2 = 1.0 - pt2 * (1.58123e-6 + c2 * (-2.9331e-8 + c2 * 1.1043e-10)) + pt2 * pt2 * 1.83e-11;
        const double dt = t2 / t - 1.0, dz = z2 / z - 1.0;
        if (!((dt < 0 ? -dt : dt) <= max_dt_rel[k])) max_dt_rel[k] = dt < 0 ? -dt : dt;
        if (!((dz < 0 ? -dz : dz) <= max_dz_rel[k])) max_dz_rel[k] = dz < 0 ? -dz : dz;
      }
      if (!a.seg(k).cubic) { // the exponent the certified evaluation hands to exp's main branch (refr_n_layer3): |e| <= 700 required
        const double e = a.seg(k).lapse != 0.0 ? a.seg(k).expo * dm_log(t / a.seg(k).tb) : a.seg(k).expo * (h - a.seg(k).hb);
        const double ae = e < 0.0 ? -e : e;
        if (!(ae <= max_abs_e[k])) max_abs_e[k] = ae;
      }
    }
  }
  return 0;
}
int ch_ray_path(const atmrt_atmosphere_t* def, const atmrt_earth_model_t* m, double wavelength, double h0, double ang_deg, int straight,
                double step, size_t n_steps, double* x, double* h) {
  AtmTableBuf buf;
  Earth e;
  if (atm_compile(*def, wavelength, buf) || earth_resolve(*m, e)) return -1;
  const AtmTable& a = buf.table();
  Stepper s;
  stepper_init(s, e.spherical != 0, e.shape_radius, h0, dm_to_radians(ang_deg));
  x[0] = 0.0; h[0] = h0;
  for (size_t k = 1; k <= n_steps; k++) {
    RayState st = stepper_next(s, a, e.spherical != 0, e.shape_radius, straight != 0, step);
    x[k] = st.x; h[k] = st.h;
  }
  return 0;
}
// one tile only: elevation + normal at points
int ch_terrain(const atmrt_earth_model_t* m, int lat0, int lon0, int n_lat, int n_lon, const int16_t* posts, size_t n, const double* lat,
               const double* lon, double* elev, int* valid, double* normal3) {
  Earth e;
  if (earth_resolve(*m, e)) return -1;
  TileDesc td{0, n_lat, n_lon};
  int32_t cell = 0;
  TerrainView tv{posts, &td, &cell, lat0, lon0, 1, 1};
  for (size_t i = 0; i < n; i++) {
    double ev = 0.0;
    valid[i] = terrain_get_elev(tv, lat[i], lon[i], ev) ? 1 : 0;
    elev[i] = ev;
    Vec3 nr = find_normal(e, tv, lat[i], lon[i]);
    normal3[3 * i] = nr.x; normal3[3 * i + 1] = nr.y; normal3[3 * i + 2] = nr.z;
  }
  return 0;
}
int ch_pixels(const atmrt_params_t* p, double* fast_dir, double* fast_elev, double* rect_dir, double* rect_elev) {
  Pinhole ph;
  pinhole_init(*p, ph);
  for (int x = 0; x < p->width; x++) fast_dir[x] = fast_ray_dir(*p, x);
  for (int y = 0; y < p->height; y++) fast_elev[y] = fast_ray_elev(*p, y);
  for (int y = 0; y < p->height; y++)
    for (int x = 0; x < p->width; x++) rect_ray_params(*p, ph, x, y, rect_dir[y * p->width + x], rect_elev[y * p->width + x]);
  return 0;
}
}

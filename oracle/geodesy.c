/* geodesy.c — EarthModel and DirectionalCalc.  ORACLE (test infrastructure).
 * Line-by-line restatement of src/utils/earth_model/mod.rs and directional_calc.rs.
 */
#include "oracle.h"
#include "oracle_math.h"

#define DEGREE_DISTANCE (10000000.0 / 90.0) /* earth_model/mod.rs:12 */
#define EARTH_R 6371000.0                   /* :14 */
#define WGS84_A 6378137.0                   /* :15 */
#define WGS84_B 6356752.314245              /* :16 */

static ovec3 v3(double x, double y, double z) { ovec3 v = {x, y, z}; return v; }
static ovec3 vscale(ovec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
static ovec3 vadd(ovec3 a, ovec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }

static int is_flat_family(int kind) {
  return kind == ATMRT_EARTH_AZIMUTHAL_EQUIDISTANT || kind == ATMRT_EARTH_FLAT_DISTORTED ||
         kind == ATMRT_EARTH_SIMPLE_OBSERVER_AE || kind == ATMRT_EARTH_OBSERVER_AE;
}

/* earth_model/mod.rs:155-172 */
static void spherical_directions(double lat, double lon, ovec3* dirn, ovec3* dire, ovec3* dirup) {
  double lat_rad = om_to_radians(lat), lon_rad = om_to_radians(lon);
  double sinlon = om_sin(lon_rad), coslon = om_cos(lon_rad);
  double sinlat = om_sin(lat_rad), coslat = om_cos(lat_rad);
  *dirup = v3(coslat * coslon, coslat * sinlon, sinlat);
  *dirn = v3(-sinlat * coslon, -sinlat * sinlon, coslat);
  *dire = v3(-sinlon, coslon, 0.0);
}

/* EarthModel::world_directions, earth_model/mod.rs:31-57 */
void oracle_world_directions(const atmrt_earth_model_t* m, double lat, double lon, ovec3* n, ovec3* e, ovec3* up) {
  if (is_flat_family(m->kind)) {
    double lon_rad = om_to_radians(lon);
    double sinlon = om_sin(lon_rad), coslon = om_cos(lon_rad);
    *n = v3(-coslon, -sinlon, 0.0);
    *e = v3(-sinlon, coslon, 0.0);
    *up = v3(0.0, 0.0, 1.0);
  } else {
    spherical_directions(lat, lon, n, e, up);
  }
}

/* spherical_to_cartesian, earth_model/mod.rs:148-153 */
static ovec3 spherical_to_cartesian(double r, double lat, double lon) {
  double x = r * om_cos(om_to_radians(lat)) * om_cos(om_to_radians(lon));
  double y = r * om_cos(om_to_radians(lat)) * om_sin(om_to_radians(lon));
  double z = r * om_sin(om_to_radians(lat));
  return v3(x, y, z);
}

static ovec3 ellipsoid_to_cartesian(double a, double b, double lat_deg, double lon_deg, double elev) {
  /* earth_model/mod.rs:73-82 */
  double e2 = 1.0 - (b * b) / (a * a);
  double lat = om_to_radians(lat_deg), lon = om_to_radians(lon_deg);
  double sl = om_sin(lat);
  double n = a / om_sqrt(1.0 - e2 * (sl * sl));
  double x = (n + elev) * om_cos(lat) * om_cos(lon);
  double y = (n + elev) * om_cos(lat) * om_sin(lon);
  double z = (n * (1.0 - e2) + elev) * om_sin(lat);
  return v3(x, y, z);
}

/* EarthModel::as_cartesian, earth_model/mod.rs:59-93 */
ovec3 oracle_as_cartesian(const atmrt_earth_model_t* m, double lat, double lon, double elev) {
  switch (m->kind) {
    case ATMRT_EARTH_SPHERICAL: return spherical_to_cartesian(m->radius + elev, lat, lon);
    case ATMRT_EARTH_SIMPLE_SPHERE: return spherical_to_cartesian(EARTH_R + elev, lat, lon);
    case ATMRT_EARTH_WGS84: return ellipsoid_to_cartesian(WGS84_A, WGS84_B, lat, lon, elev);
    case ATMRT_EARTH_ELLIPSOID: return ellipsoid_to_cartesian(m->a, m->b, lat, lon, elev);
    default: {
      double z = elev;
      double r = (90.0 - lat) * DEGREE_DISTANCE;
      double x = r * om_cos(om_to_radians(lon));
      double y = r * om_sin(om_to_radians(lon));
      return v3(x, y, z);
    }
  }
}

/* EarthModel::to_shape, earth_model/mod.rs:95-112 */
int oracle_to_shape(const atmrt_earth_model_t* m, double* radius) {
  switch (m->kind) {
    case ATMRT_EARTH_SIMPLE_SPHERE: *radius = EARTH_R; return 1;
    case ATMRT_EARTH_SPHERICAL: *radius = m->radius; return 1;
    case ATMRT_EARTH_WGS84: *radius = (2.0 * WGS84_A + WGS84_B) / 3.0; return 1;
    case ATMRT_EARTH_ELLIPSOID: *radius = (2.0 * m->a + m->b) / 3.0; return 1;
    default: *radius = 0.0; return 0;
  }
}

/* SphericalCalc::new, directional_calc.rs:56-69 */
static void spherical_calc_new(double radius, double lat, double lon, double dir, oracle_dircalc* c) {
  ovec3 dirn, dire, pos;
  double dir_rad, sindir, cosdir;
  spherical_directions(lat, lon, &dirn, &dire, &pos);
  dir_rad = om_to_radians(dir);
  sindir = om_sin(dir_rad);
  cosdir = om_cos(dir_rad);
  c->kind = 2;
  c->radius = radius;
  c->pos = pos;
  c->dir = vadd(vscale(dirn, cosdir), vscale(dire, sindir));
}

/* EllipsoidCalc::new, directional_calc.rs:103-131 */
static void ellipsoid_calc_new(double a, double b, double lat_deg, double lon_deg, double dir, oracle_dircalc* c) {
  double lat = om_to_radians(lat_deg), lon = om_to_radians(lon_deg), az1 = om_to_radians(dir);
  double f = (a - b) / a;
  double red_lat = om_atan((1.0 - f) * om_tan(lat));
  double sig1 = om_atan(om_tan(red_lat) / om_cos(az1));
  double alfa = om_asin(om_cos(red_lat) * om_sin(az1));
  double ca = om_cos(alfa);
  double u2 = ca * ca * (a * a - b * b) / (b * b);
  c->kind = 3;
  c->cap_a = 1.0 + u2 / 256.0 * (64.0 + u2 * (-12.0 + 5.0 * u2));
  c->cap_b = u2 / 512.0 * (128.0 + u2 * (-64.0 + 37.0 * u2));
  c->cap_c = f / 16.0 * (ca * ca) * (4.0 + f * (4.0 - 3.0 * (ca * ca)));
  c->b = b;
  c->f = f;
  c->red_lat = red_lat;
  c->lon = lon;
  c->az1 = az1;
  c->alfa = alfa;
  c->sig1 = sig1;
}

/* EarthModel::coords_at_dist_calc, earth_model/mod.rs:114-145 */
void oracle_dircalc_new(const atmrt_earth_model_t* m, double lat, double lon, double dir, oracle_dircalc* c) {
  switch (m->kind) {
    case ATMRT_EARTH_AZIMUTHAL_EQUIDISTANT: {
      ovec3 vn, ve, vu;
      double dr = om_to_radians(dir);
      c->kind = 0;
      c->pos = oracle_as_cartesian(m, lat, lon, 0.0);
      oracle_world_directions(m, lat, lon, &vn, &ve, &vu);
      c->dir = vadd(vscale(vn, om_cos(dr)), vscale(ve, om_sin(dr)));
      break;
    }
    case ATMRT_EARTH_FLAT_DISTORTED:
      c->kind = 1;
      c->start_lat = lat;
      c->start_lon = lon;
      c->dir_deg = dir;
      break;
    case ATMRT_EARTH_OBSERVER_AE:
    case ATMRT_EARTH_SPHERICAL: spherical_calc_new(m->radius, lat, lon, dir, c); break;
    case ATMRT_EARTH_SIMPLE_SPHERE:
    case ATMRT_EARTH_SIMPLE_OBSERVER_AE: spherical_calc_new(EARTH_R, lat, lon, dir, c); break;
    case ATMRT_EARTH_ELLIPSOID: ellipsoid_calc_new(m->a, m->b, lat, lon, dir, c); break;
    default: ellipsoid_calc_new(WGS84_A, WGS84_B, lat, lon, dir, c); break; /* Wgs84 */
  }
}

/* DirectionalCalc::coords_at_dist for the four implementors */
void oracle_coords_at_dist(const oracle_dircalc* c, double dist, double* lat, double* lon) {
  switch (c->kind) {
    case 0: { /* AzEqCalc, directional_calc.rs:20-28 */
      double px = c->pos.x + c->dir.x * dist, py = c->pos.y + c->dir.y * dist;
      double r;
      *lon = om_to_degrees(om_atan2(py, px));
      r = om_sqrt(px * px + py * py);
      *lat = 90.0 - r / DEGREE_DISTANCE;
      break;
    }
    case 1: { /* FlDsCalc, directional_calc.rs:41-48 */
      double d_lat = om_cos(om_to_radians(c->dir_deg)) * dist / DEGREE_DISTANCE;
      double d_lon = om_sin(om_to_radians(c->dir_deg)) * dist / DEGREE_DISTANCE / om_cos(om_to_radians(c->start_lat));
      *lat = c->start_lat + d_lat;
      *lon = c->start_lon + d_lon;
      break;
    }
    case 2: { /* SphericalCalc, directional_calc.rs:72-85 */
      double ang = dist / c->radius;
      double sinang = om_sin(ang), cosang = om_cos(ang);
      double fx = c->pos.x * cosang + c->dir.x * sinang;
      double fy = c->pos.y * cosang + c->dir.y * sinang;
      double fz = c->pos.z * cosang + c->dir.z * sinang;
      *lat = om_to_degrees(om_asin(fz));
      *lon = om_to_degrees(om_atan2(fy, fx));
      break;
    }
    default: { /* EllipsoidCalc (Vincenty direct), directional_calc.rs:135-185 */
      double sig = dist / c->b / c->cap_a, sigm, lat2, lambda, dl, t, cs;
      int it;
      /* the reference loops until |d sigma| < 1e-10 with no bound (and would spin on NaN);
       * the bound of 64 is never reached for finite input (contraction factor ~ cap_b ~ 1e-3). */
      for (it = 0; it < 64; it++) {
        double dsig, new_sig;
        sigm = 2.0 * c->sig1 + sig;
        cs = om_cos(sigm);
        dsig = c->cap_b * om_sin(sig) * (cs + c->cap_b / 4.0 * om_cos(sig) * (-1.0 + 2.0 * (cs * cs)));
        new_sig = dist / c->b / c->cap_a + dsig;
        dsig = om_fabs(new_sig - sig);
        sig = new_sig;
        if (dsig < 1e-10) break;
      }
      sigm = 2.0 * c->sig1 + sig;
      t = om_sin(c->red_lat) * om_sin(sig) - om_cos(c->red_lat) * om_cos(sig) * om_cos(c->az1);
      {
        double sa = om_sin(c->alfa);
        lat2 = om_atan((om_sin(c->red_lat) * om_cos(sig) + om_cos(c->red_lat) * om_sin(sig) * om_cos(c->az1)) /
                       ((1.0 - c->f) * om_sqrt(sa * sa + t * t)));
      }
      lambda = om_atan(om_sin(sig) * om_sin(c->az1) /
                       (om_cos(c->red_lat) * om_cos(sig) - om_sin(c->red_lat) * om_sin(sig) * om_cos(c->az1)));
      cs = om_cos(sigm);
      dl = lambda - (1.0 - c->cap_c) * c->f * om_sin(c->alfa) *
                        (sig + c->cap_c * om_sin(sig) * (cs + c->cap_c * om_cos(sig) * (-1.0 + 2.0 * (cs * cs))));
      *lat = om_to_degrees(lat2);
      *lon = om_to_degrees(c->lon + dl);
      break;
    }
  }
}

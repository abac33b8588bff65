/* objects.c — scene objects: frustum / billboard intersection and proximity filter.
 * ORACLE (test infrastructure).  Line-by-line restatement of src/object/{mod,frustum,billboard}.rs.
 */
#include "oracle_internal.h"

static ovec3 v3(double x, double y, double z) { ovec3 v = {x, y, z}; return v; }
static ovec3 vsub(ovec3 a, ovec3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
static ovec3 vadd(ovec3 a, ovec3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
static ovec3 vscale(ovec3 a, double s) { return v3(a.x * s, a.y * s, a.z * s); }
static ovec3 vdiv(ovec3 a, double s) { return v3(a.x / s, a.y / s, a.z / s); }
static ovec3 vneg(ovec3 a) { return v3(-a.x, -a.y, -a.z); }
static double vdot(ovec3 a, ovec3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static ovec3 vcross(ovec3 a, ovec3 b) {
  return v3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static int in_range(double lo, double x, double hi) { return lo <= x && x < hi; } /* (lo..hi).contains(&x) */

static void push(oracle_collision* out, int* n, double prop, ovec3 normal, const double color[4]) {
  int i = *n, k;
  /* results.sort_by(prop) is stable: insert after every element with prop <= new prop */
  while (i > 0 && out[i - 1].prop > prop) {
    out[i] = out[i - 1];
    i--;
  }
  out[i].prop = prop;
  out[i].normal = normal;
  for (k = 0; k < 4; k++) out[i].color[k] = color[k];
  (*n)++;
}

/* Frustum::check_collision, frustum.rs:18-101.  Returns the number of collisions (<= 4), sorted by prop. */
static int frustum_collision(const oracle_object* o, const atmrt_earth_model_t* m, ocoords point1, ocoords point2,
                             oracle_collision* out) {
  ovec3 pos1 = oracle_as_cartesian(m, point1.lat, point1.lon, point1.elev);
  ovec3 pos2 = oracle_as_cartesian(m, point2.lat, point2.lon, point2.elev);
  ovec3 obj_pos = oracle_as_cartesian(m, o->lat, o->lon, o->elev);
  ovec3 p1 = vsub(pos1, obj_pos);
  double p1sq = vdot(p1, p1);
  ovec3 dn, de, v, w;
  double wsq, p1v, p1w, wv, aa, aa1, a, b, c, delta;
  int n = 0, side;
  oracle_collision unsorted[4];
  int nu = 0, i;
  oracle_world_directions(m, o->lat, o->lon, &dn, &de, &v);
  w = vsub(pos2, pos1);
  wsq = vdot(w, w);
  p1v = vdot(p1, v);
  p1w = vdot(p1, w);
  wv = vdot(w, v);
  aa = (o->r2 - o->r1) / o->height;
  aa1 = 1.0 + aa * aa;
  a = wsq - wv * wv * (1.0 + aa * aa);
  b = 2.0 * (p1w - wv * (p1v * aa1 + aa * o->r1));
  c = p1sq - p1v * p1v * aa1 - o->r1 * o->r1 - 2.0 * aa * o->r1 * p1v;
  delta = b * b - 4.0 * a * c;
  /* side surface */
  if (delta >= 0.0) {
    double x1 = (-b - om_sqrt(delta)) / 2.0 / a;
    double x2 = (-b + om_sqrt(delta)) / 2.0 / a;
    double tmp[2];
    int nt = 0;
    if (a < 0.0) {
      double t = x1;
      x1 = x2;
      x2 = t;
    }
    if (in_range(0.0, x1, 1.0)) tmp[nt++] = x1;
    if (in_range(0.0, x2, 1.0)) tmp[nt++] = x2;
    for (i = 0; i < nt; i++) {
      double x = tmp[i];
      ovec3 intersection = vadd(p1, vscale(w, x));
      double h = vdot(intersection, v);
      ovec3 outward, normal;
      double o_len, ang;
      if (!in_range(0.0, h, o->height)) continue;
      outward = vsub(intersection, vscale(v, h));
      o_len = om_sqrt(vdot(outward, outward));
      outward = vdiv(outward, o_len);
      ang = om_atan2(o->r1 - o->r2, o->height);
      normal = vadd(vscale(outward, om_cos(ang)), vscale(v, om_sin(ang)));
      unsorted[nu].prop = x;
      unsorted[nu].normal = normal;
      nu++;
    }
  }
  /* top and bottom */
  for (side = 0; side < 2; side++) {
    double h = side ? o->height : 0.0, r = side ? o->r2 : o->r1;
    ovec3 nrm = side ? v : vneg(v);
    double x = (h - p1v) / wv;
    ovec3 outv = vsub(vadd(p1, vscale(w, x)), vscale(v, h));
    double d = vdot(outv, outv);
    if (d < r * r && in_range(0.0, x, 1.0)) {
      unsorted[nu].prop = x;
      unsorted[nu].normal = nrm;
      nu++;
    }
  }
  for (i = 0; i < nu; i++) push(out, &n, unsorted[i].prop, unsorted[i].normal, o->color);
  return n;
}

/* Image::get_pixel, object/mod.rs:91-117, followed by the /255 of billboard.rs:58-63 */
static void texture_fetch(const oracle_object* o, double x, double y, double color[4]) {
  double w = (double)o->tex_w, h = (double)o->tex_h;
  double x1, x2, y1, y2, px, py;
  unsigned ix1, ix2, iy1, iy2;
  int ch;
  x = x * w - 0.5;
  x1 = om_floor(x);
  if (x1 < 0.0) x1 = 0.0;
  if (x1 > w - 2.0) x1 = w - 2.0;
  x2 = x1 + 1.0;
  ix1 = (unsigned)x1;
  ix2 = (unsigned)x2;
  y = (1.0 - y) * h - 0.5;
  y1 = om_floor(y);
  if (y1 < 0.0) y1 = 0.0;
  if (y1 > h - 2.0) y1 = h - 2.0;
  y2 = y1 + 1.0;
  iy1 = (unsigned)y1;
  iy2 = (unsigned)y2;
  px = x - x1;
  py = y - y1;
  for (ch = 0; ch < 4; ch++) {
    double p00 = (double)o->tex[((size_t)iy1 * o->tex_w + ix1) * 4 + ch] / 255.0;
    double p01 = (double)o->tex[((size_t)iy2 * o->tex_w + ix1) * 4 + ch] / 255.0;
    double p10 = (double)o->tex[((size_t)iy1 * o->tex_w + ix2) * 4 + ch] / 255.0;
    double p11 = (double)o->tex[((size_t)iy2 * o->tex_w + ix2) * 4 + ch] / 255.0;
    double v = p00 * (1.0 - px) * (1.0 - py) + p01 * (1.0 - px) * py + p10 * px * (1.0 - py) + p11 * px * py;
    double q = v * 255.0; /* vec4_to_rgba: `as u8` truncates and saturates (utils/mod.rs:41-47) */
    unsigned u = q != q ? 0u : q <= 0.0 ? 0u : q >= 255.0 ? 255u : (unsigned)q;
    color[ch] = (double)u / 255.0;
  }
}

/* Billboard::check_collision, billboard.rs:17-66 */
static int billboard_collision(const oracle_object* o, const atmrt_earth_model_t* m, ocoords point1, ocoords point2,
                               oracle_collision* out) {
  ovec3 pos1 = oracle_as_cartesian(m, point1.lat, point1.lon, point1.elev);
  ovec3 pos2 = oracle_as_cartesian(m, point2.lat, point2.lon, point2.elev);
  ovec3 obj_pos = oracle_as_cartesian(m, o->lat, o->lon, o->elev);
  ovec3 ray = vsub(pos2, pos1);
  ovec3 dn, de, up, right, front, p1, intersection;
  double right_len, prop, x, y;
  int k;
  oracle_world_directions(m, o->lat, o->lon, &dn, &de, &up);
  right = vcross(ray, up);
  right_len = om_sqrt(vdot(right, right));
  right = vdiv(right, right_len);
  front = vcross(right, up);
  p1 = vsub(pos1, obj_pos);
  prop = -vdot(p1, front) / vdot(ray, front);
  if (!in_range(0.0, prop, 1.0)) return 0;
  intersection = vadd(p1, vscale(ray, prop));
  y = vdot(intersection, up);
  x = vdot(intersection, right);
  if (!in_range(0.0, y, o->height) || !in_range(-o->width / 2.0, x, o->width / 2.0)) return 0;
  x = (x + o->width / 2.0) / o->width;
  y = y / o->height;
  out[0].prop = prop;
  out[0].normal = front;
  texture_fetch(o, x, y, out[0].color);
  (void)k;
  return 1;
}

int oracle_object_collision(const oracle_object* o, const atmrt_earth_model_t* m, ocoords p1, ocoords p2,
                            oracle_collision* out) {
  return o->kind == ATMRT_OBJ_FRUSTUM ? frustum_collision(o, m, p1, p2, out) : billboard_collision(o, m, p1, p2, out);
}

/* Object::is_close, frustum.rs:103-114 and billboard.rs:68-78 */
int oracle_object_is_close(const oracle_object* o, const atmrt_earth_model_t* m, double sim_step, double lat,
                           double lon) {
  ovec3 obj_pos = oracle_as_cartesian(m, o->lat, o->lon, o->elev);
  ovec3 pos = oracle_as_cartesian(m, lat, lon, o->elev);
  ovec3 d = vsub(pos, obj_pos);
  double r = o->kind == ATMRT_OBJ_FRUSTUM ? (o->r1 > o->r2 ? o->r1 : o->r2) : o->width; /* f64::max */
  return vdot(d, d) < 2.0 * (r + sim_step) * (r + sim_step);
}

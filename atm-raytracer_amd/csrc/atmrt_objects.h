// atmrt_objects.h — scene objects on the device: proximity filter and ray-segment intersection.
// Restates src/object/frustum.rs, billboard.rs and mod.rs:91-117 in the operation order of the CPU
// checker (oracle/objects.c), so object hits are bit-identical too.
#pragma once

#include "atmrt_core.h"

namespace atmrt {

// SerializableObject (object/mod.rs:185-190) resolved on the device by k_resolve.
struct ObjectDev {
  int32_t kind; // atmrt_object_kind
  int32_t tex_w, tex_h;
  int32_t _pad;
  double lat, lon, elev;  // elev: Altitude::abs (object/mod.rs:166-175)
  double r1, r2, height, width;
  double color[4];
  int64_t tex_offset;     // first byte of the RGBA8 texture in the texture pool
  // derived once per frame (pure functions of the fields above):
  Vec3 pos;               // earth_model.as_cartesian(&self.position)
  Vec3 up;                // earth_model.world_directions(lat, lon).2
  double close2;          // 2 (r + sim_step)^2 of Object::is_close
  double vlo, vhi;        // a ray segment whose elevations lie entirely below vlo or above vhi cannot touch the object
};

// sin/cos of a sample's latitude and longitude, shared by every as_cartesian of that sample
struct LatLonTrig {
  double sl, cl, so, co;
  double lat;
};
ATMRT_HD LatLonTrig latlon_trig(const Earth& e, double lat, double lon) {
  LatLonTrig t;
  t.lat = lat;
  dm_sincos(dm_to_radians(lon), &t.so, &t.co);
  if (e.cart != 0) dm_sincos(dm_to_radians(lat), &t.sl, &t.cl);
  else t.sl = t.cl = 0.0;
  return t;
}
// EarthModel::as_cartesian (mod.rs:59-93) for a point whose trigonometry is already known
ATMRT_HD Vec3 as_cartesian_trig(const Earth& e, const LatLonTrig& t, double elev) {
  if (e.cart == 1) {
    double r = e.cart_radius + elev;
    return v3(r * t.cl * t.co, r * t.cl * t.so, r * t.sl);
  }
  if (e.cart == 2) {
    double e2 = 1.0 - (e.b * e.b) / (e.a * e.a);
    double n = e.a / dm_sqrt(1.0 - e2 * (t.sl * t.sl));
    return v3((n + elev) * t.cl * t.co, (n + elev) * t.cl * t.so, (n * (1.0 - e2) + elev) * t.sl);
  }
  double r = (90.0 - t.lat) * DEGREE_DISTANCE;
  return v3(r * t.co, r * t.so, elev);
}

ATMRT_HD void object_derive(const Earth& e, double sim_step, ObjectDev& o) {
  o.pos = as_cartesian(e, o.lat, o.lon, o.elev);
  Vec3 n, ea;
  world_directions(e, o.lat, o.lon, n, ea, o.up);
  double r = o.kind == ATMRT_OBJ_FRUSTUM ? (o.r1 > o.r2 ? o.r1 : o.r2) : o.width; // frustum.rs:111, billboard.rs:77
  o.close2 = 2.0 * (r + sim_step) * (r + sim_step);
  // Every collision of frustum.rs / billboard.rs has its coordinate along `up` in [0, height].  For a point of a segment
  // between two samples that coordinate is its elevation minus o.elev, less the curvature drop over its horizontal offset
  // (<= reach^2 / 2R; only close objects are tested, so the offset is below reach) and give or take the chord's sagitta
  // (step^2 / 8R).  Both are doubled here and 2 m added: segments outside [vlo, vhi] are skipped before any geometry.
  double rr = e.cart == 1 ? e.cart_radius : e.cart == 2 ? (e.a < e.b ? e.a : e.b) : 0.0;
  double reach = dm_sqrt(o.close2) + sim_step;
  double curv = e.cart == 0 ? 0.0 : (sim_step * sim_step / 4.0 + reach * reach) / rr;
  o.vlo = o.elev - 2.0 - curv;
  o.vhi = o.elev + (o.height > 0.0 ? o.height : 0.0) + 2.0 + curv;
}

// Object::is_close, frustum.rs:103-114 / billboard.rs:68-78
ATMRT_HD bool object_is_close(const Earth& e, const ObjectDev& o, const LatLonTrig& t) {
  Vec3 d = as_cartesian_trig(e, t, o.elev) - o.pos;
  return dot(d, d) < o.close2;
}

struct Collision {
  double prop;
  Vec3 normal;
  double color[4];
};

ATMRT_HD bool in_range(double lo, double x, double hi) { return lo <= x && x < hi; } // (lo..hi).contains(&x)

// Image::get_pixel (object/mod.rs:91-117) + the /255 of billboard.rs:58-63
ATMRT_HD void texture_fetch(const ObjectDev& o, const uint8_t* textures, double x, double y, double* color) {
  const uint8_t* tex = textures + o.tex_offset;
  double w = (double)o.tex_w, h = (double)o.tex_h;
  x = x * w - 0.5;
  double x1 = dm_floor(x);
  if (x1 < 0.0) x1 = 0.0;
  if (x1 > w - 2.0) x1 = w - 2.0;
  double x2 = x1 + 1.0;
  unsigned ix1 = (unsigned)x1, ix2 = (unsigned)x2;
  y = (1.0 - y) * h - 0.5;
  double y1 = dm_floor(y);
  if (y1 < 0.0) y1 = 0.0;
  if (y1 > h - 2.0) y1 = h - 2.0;
  double y2 = y1 + 1.0;
  unsigned iy1 = (unsigned)y1, iy2 = (unsigned)y2;
  double px = x - x1, py = y - y1;
  for (int ch = 0; ch < 4; ch++) {
    double p00 = (double)tex[((size_t)iy1 * o.tex_w + ix1) * 4 + ch] / 255.0;
    double p01 = (double)tex[((size_t)iy2 * o.tex_w + ix1) * 4 + ch] / 255.0;
    double p10 = (double)tex[((size_t)iy1 * o.tex_w + ix2) * 4 + ch] / 255.0;
    double p11 = (double)tex[((size_t)iy2 * o.tex_w + ix2) * 4 + ch] / 255.0;
    double v = p00 * (1.0 - px) * (1.0 - py) + p01 * (1.0 - px) * py + p10 * px * (1.0 - py) + p11 * px * py;
    double q = v * 255.0; // vec4_to_rgba: `as u8` truncates and saturates (utils/mod.rs:41-47)
    unsigned u = q != q ? 0u : q <= 0.0 ? 0u : q >= 255.0 ? 255u : (unsigned)q;
    color[ch] = (double)u / 255.0;
  }
}

// Object::check_collision.  pos1/pos2: cartesian ends of the ray segment.  Returns the number of
// collisions (<= 4) written to out[], sorted by prop (stable, like results.sort_by in frustum.rs:98).
ATMRT_HD int object_collision(const ObjectDev& o, const uint8_t* textures, Vec3 pos1, Vec3 pos2, Collision* out) {
  if (o.kind == ATMRT_OBJ_FRUSTUM) { // frustum.rs:18-101
    Vec3 p1 = pos1 - o.pos;
    double p1sq = dot(p1, p1);
    Vec3 v = o.up;
    Vec3 w = pos2 - pos1;
    double wsq = dot(w, w), p1v = dot(p1, v), p1w = dot(p1, w), wv = dot(w, v);
    double aa = (o.r2 - o.r1) / o.height;
    double aa1 = 1.0 + aa * aa;
    double a = wsq - wv * wv * (1.0 + aa * aa);
    double b = 2.0 * (p1w - wv * (p1v * aa1 + aa * o.r1));
    double c = p1sq - p1v * p1v * aa1 - o.r1 * o.r1 - 2.0 * aa * o.r1 * p1v;
    double delta = b * b - 4.0 * a * c;
    double props[4];
    Vec3 normals[4];
    int nu = 0;
    if (delta >= 0.0) {
      double x1 = (-b - dm_sqrt(delta)) / 2.0 / a;
      double x2 = (-b + dm_sqrt(delta)) / 2.0 / a;
      if (a < 0.0) {
        double t = x1;
        x1 = x2;
        x2 = t;
      }
      for (int s = 0; s < 2; s++) {
        double x = s == 0 ? x1 : x2;
        if (!in_range(0.0, x, 1.0)) continue;
        Vec3 intersection = p1 + w * x;
        double h = dot(intersection, v);
        if (!in_range(0.0, h, o.height)) continue;
        Vec3 outward = intersection - v * h;
        double o_len = dm_sqrt(dot(outward, outward));
        outward = outward / o_len;
        double ang = dm_atan2(o.r1 - o.r2, o.height);
        props[nu] = x;
        normals[nu] = outward * dm_cos(ang) + v * dm_sin(ang);
        nu++;
      }
    }
    for (int side = 0; side < 2; side++) { // top and bottom, frustum.rs:88-96
      double h = side ? o.height : 0.0, r = side ? o.r2 : o.r1;
      double x = (h - p1v) / wv;
      Vec3 outv = (p1 + w * x) - v * h;
      double d = dot(outv, outv);
      if (d < r * r && in_range(0.0, x, 1.0)) {
        props[nu] = x;
        normals[nu] = side ? v : -v;
        nu++;
      }
    }
    int n = 0;
    for (int i = 0; i < nu; i++) { // stable insertion
      int j = n;
      while (j > 0 && out[j - 1].prop > props[i]) {
        out[j] = out[j - 1];
        j--;
      }
      out[j].prop = props[i];
      out[j].normal = normals[i];
      for (int k = 0; k < 4; k++) out[j].color[k] = o.color[k];
      n++;
    }
    return n;
  }
  // Billboard::check_collision, billboard.rs:17-66
  Vec3 ray = pos2 - pos1;
  Vec3 up = o.up;
  Vec3 right = cross(ray, up);
  double right_len = dm_sqrt(dot(right, right));
  right = right / right_len;
  Vec3 front = cross(right, up);
  Vec3 p1 = pos1 - o.pos;
  double prop = -dot(p1, front) / dot(ray, front);
  if (!in_range(0.0, prop, 1.0)) return 0;
  Vec3 intersection = p1 + ray * prop;
  double y = dot(intersection, up);
  double x = dot(intersection, right);
  if (!in_range(0.0, y, o.height) || !in_range(-o.width / 2.0, x, o.width / 2.0)) return 0;
  x = (x + o.width / 2.0) / o.width;
  y = y / o.height;
  out[0].prop = prop;
  out[0].normal = front;
  texture_fetch(o, textures, x, y, out[0].color);
  return 1;
}

} // namespace atmrt

// atmrt_render.h — renderer compositing + colouring in the operation order of the reference:
// src/renderer/mod.rs:367-414 (fog, add, draw_image), src/coloring/{simple,shading}.rs, src/utils/mod.rs:16-29,
// ConfColoring::into_coloring (params.rs:231-277).  u8 conversions are Rust's `as u8` (truncate, saturate, NaN -> 0).
#pragma once

#include "atmrt_core.h"

namespace atmrt {

ATMRT_HD uint8_t as_u8(double v) {
  if (v != v || v <= 0.0) return 0;
  if (v >= 255.0) return 255;
  return (uint8_t)v;
}

// Rust `a % b` (fmod) for the magnitudes the renderer produces (|a / b| < 2^20): exact
ATMRT_HD double fmod_small(double a, double b) {
  if (!(dm_fabs(a) < 1048576.0 * dm_fabs(b))) return a - a;
  double q = a / b;
  q = q < 0.0 ? -dm_floor(-q) : dm_floor(q);
  double r = a - q * b;
  if (a >= 0.0 && r < 0.0) r += dm_fabs(b);
  if (a < 0.0 && r > 0.0) r -= dm_fabs(b);
  return r;
}

struct Rgb8 {
  uint8_t c[3];
};

// hsv, coloring/simple.rs:57-87
ATMRT_HD Rgb8 hsv(double h, double s, double v) {
  double c = v * s;
  double hm = fmod_small(h, 360.0);
  h = hm < 0.0 ? hm + 360.0 : hm;
  double x = c * (1.0 - dm_fabs(fmod_small(h / 60.0, 2.0) - 1.0));
  double m = v - c;
  double rp, gp, bp;
  if (h >= 0.0 && h < 60.0) { rp = c; gp = x; bp = 0.0; }
  else if (h >= 60.0 && h < 120.0) { rp = x; gp = c; bp = 0.0; }
  else if (h >= 120.0 && h < 180.0) { rp = 0.0; gp = c; bp = x; }
  else if (h >= 180.0 && h < 240.0) { rp = 0.0; gp = x; bp = c; }
  else if (h >= 240.0 && h < 300.0) { rp = x; gp = 0.0; bp = c; }
  else { rp = c; gp = 0.0; bp = x; }
  return Rgb8{{as_u8((rp + m) * 255.0), as_u8((gp + m) * 255.0), as_u8((bp + m) * 255.0)}};
}

// SimpleColors::color_for_pixel, coloring/simple.rs:22-45
ATMRT_HD Rgb8 simple_color(const atmrt_coloring_t& c, double distance, double elevation) {
  double dist_ratio = distance / c.max_distance;
  if (elevation <= c.water_level) {
    double mul = 1.0 - dist_ratio * 0.6;
    return Rgb8{{0, as_u8(128.0 * mul), as_u8(255.0 * mul)}};
  }
  double elev_ratio = elevation / 4500.0;
  double h = 120.0 - 240.0 * (elev_ratio < 0.0 ? -dm_pow(-elev_ratio, 0.65) : dm_pow(elev_ratio, 0.65));
  double v = (elev_ratio > 0.7 ? 2.1 - elev_ratio * 2.0 : 0.9 - elev_ratio / 0.7 * 0.2) * (1.0 - dist_ratio * 0.6);
  double s = 1.0 - dist_ratio * 0.9;
  return hsv(h, s, v);
}

// ColorPalette, coloring/shading.rs:16-83
ATMRT_HD void palette_elev(int palette, double elev, double* out) {
  const bool legacy = palette == ATMRT_PALETTE_LEGACY;
  const double c0[3] = {legacy ? 0.0 : 0.4, legacy ? 1.0 : 0.8, legacy ? 0.0 : 0.3};
  const double c1[3] = {legacy ? 0.6 : 0.77, legacy ? 1.0 : 0.84, legacy ? 0.0 : 0.4};
  const double c2[3] = {legacy ? 0.5 : 0.41, legacy ? 0.5 : 0.52, legacy ? 0.5 : 0.4};
  const double c3[3] = {legacy ? 1.0 : 0.85, legacy ? 1.0 : 0.92, legacy ? 1.0 : 0.95};
  const double t1 = 300.0, t2 = legacy ? 1200.0 : 1000.0, t3 = 1800.0, t4 = 3000.0;
  for (int i = 0; i < 3; i++) {
    if (elev < t1) out[i] = c0[i];
    else if (elev < t2) { double prop = (elev - t1) / (t2 - t1); out[i] = c1[i] * prop + c0[i] * (1.0 - prop); }
    else if (elev < t3) { double prop = (elev - t2) / (t3 - t2); out[i] = c2[i] * prop + c1[i] * (1.0 - prop); }
    else if (elev < t4) { double prop = (elev - t3) / (t4 - t3); out[i] = c3[i] * prop + c2[i] * (1.0 - prop); }
    else out[i] = c3[i];
  }
}

// Shading::color_for_pixel, coloring/shading.rs:111-137
ATMRT_HD Rgb8 shading_color(const atmrt_coloring_t& c, double nx, double ny, double nz, double elevation, uint32_t tag,
                            double r, double g, double b) {
  double light_dot = c.light_dir[0] * nx + c.light_dir[1] * ny + c.light_dir[2] * nz;
  light_dot = light_dot >= 0.0 ? light_dot : 0.0;
  double brightness = c.ambient_light + (1.0 - c.ambient_light) * light_dot * light_dot;
  double col[3];
  if (tag == ATMRT_COLOR_RGBA) {
    col[0] = r; col[1] = g; col[2] = b;
  } else if (elevation <= c.water_level) {
    const bool legacy = c.palette == ATMRT_PALETTE_LEGACY;
    col[0] = legacy ? 0.0 : 0.23; col[1] = legacy ? 0.5 : 0.41; col[2] = legacy ? 1.0 : 0.55;
  } else {
    palette_elev(c.palette, elevation, col);
  }
  return Rgb8{{as_u8(col[0] * brightness * 255.0), as_u8(col[1] * brightness * 255.0), as_u8(col[2] * brightness * 255.0)}};
}

ATMRT_HD Rgb8 default_color(const atmrt_coloring_t& c) { // draw_image :388-394
  if (c.has_fog) return Rgb8{{160, 160, 160}};
  if (c.kind == ATMRT_COLORING_SIMPLE) return Rgb8{{28, 28, 28}};
  const bool legacy = c.palette == ATMRT_PALETTE_LEGACY;
  return Rgb8{{as_u8((legacy ? 0.11 : 0.23) * 255.0), as_u8((legacy ? 0.11 : 0.41) * 255.0), as_u8((legacy ? 0.11 : 0.55) * 255.0)}};
}

// fog, renderer/mod.rs:367-376
ATMRT_HD Rgb8 apply_fog(double fog_dist, double pixel_dist, Rgb8 color) {
  double fog_coeff = 1.0 - dm_exp(-pixel_dist / fog_dist);
  Rgb8 o;
  for (int i = 0; i < 3; i++) o.c[i] = as_u8((double)color.c[i] * (1.0 - fog_coeff) + 160.0 * fog_coeff);
  return o;
}
// add, renderer/mod.rs:378-383
ATMRT_HD Rgb8 add_rgb(Rgb8 acc, Rgb8 c2, double a) {
  Rgb8 o;
  for (int i = 0; i < 3; i++) o.c[i] = as_u8(((double)acc.c[i] / 255.0 + (double)c2.c[i] / 255.0 * a) * 255.0);
  return o;
}

// ConfColoring::into_coloring, params.rs:231-277
ATMRT_HD int coloring_from_conf(const atmrt_params_t& p, int32_t kind, double water_level, double ambient_light,
                                double light_zenith_angle, double light_dir, int32_t palette, int32_t has_fog, double fog_distance,
                                atmrt_coloring_t& out) {
  out.kind = kind;
  out.palette = palette;
  out.water_level = water_level;
  out.max_distance = p.frame.max_distance;
  out.ambient_light = ambient_light;
  out.light_dir[0] = out.light_dir[1] = out.light_dir[2] = 0.0;
  out.has_fog = has_fog;
  out._pad = 0;
  out.fog_distance = fog_distance;
  if (kind != ATMRT_COLORING_SIMPLE && kind != ATMRT_COLORING_SHADING) return -1;
  if (kind == ATMRT_COLORING_SHADING) {
    Earth e;
    if (earth_resolve(p.earth, e)) return -1;
    double lza = dm_to_radians(light_zenith_angle), ld = dm_to_radians(light_dir), fa = dm_to_radians(p.frame.direction);
    Vec3 n, ea, u;
    world_directions(e, p.position.latitude, p.position.longitude, n, ea, u);
    Vec3 front = n * dm_cos(fa) + ea * dm_sin(fa);
    Vec3 right = ea * dm_cos(fa) - n * dm_sin(fa);
    Vec3 v = (-front) * dm_sin(lza) * dm_cos(ld) + right * dm_sin(lza) * dm_sin(ld) + u * dm_cos(lza);
    double len = dm_sqrt(v.x * v.x + v.y * v.y + v.z * v.z);
    out.light_dir[0] = v.x / len;
    out.light_dir[1] = v.y / len;
    out.light_dir[2] = v.z / len;
  }
  return 0;
}

} // namespace atmrt

/* render.c — renderer compositing + colouring.  ORACLE (test infrastructure).
 * Line-by-line restatement of src/renderer/mod.rs:367-414 (fog, add, draw_image), src/coloring/{simple,shading}.rs,
 * src/utils/mod.rs:16-29 (rgb <-> vec3 with `as u8` truncation) and ConfColoring::into_coloring (params.rs:231-277).
 * Everything here is in the reference repository, so this part of the oracle is pinned to source.
 */
#include "oracle_internal.h"

#include <math.h>
#include <string.h>

/* Rust `f as u8`: truncates toward zero, saturates, NaN -> 0 */
static uint8_t as_u8(double v) {
  if (v != v || v <= 0.0) return 0;
  if (v >= 255.0) return 255;
  return (uint8_t)v;
}

/* Rust `a % b` on f64 is C fmod.  det flavour: exact for |a / b| < 2^20 by peeling the truncated quotient. */
static double om_fmod(double a, double b) {
#ifdef ORACLE_LIBM
  return fmod(a, b);
#else
  double q, r;
  if (!(om_fabs(a) < 1048576.0 * om_fabs(b))) return a - a; /* outside the range the renderer produces */
  q = a / b;
  q = q < 0.0 ? -om_floor(-q) : om_floor(q);
  r = a - q * b; /* exact: q*b has few bits and cancels against a */
  if (a >= 0.0 && r < 0.0) r += om_fabs(b);
  if (a < 0.0 && r > 0.0) r -= om_fabs(b);
  return r;
#endif
}

/* hsv, coloring/simple.rs:57-87 */
static void hsv(double h, double s, double v, uint8_t rgb[3]) {
  double c = v * s, x, m, rp, gp, bp;
  h = om_fmod(h, 360.0) < 0.0 ? om_fmod(h, 360.0) + 360.0 : om_fmod(h, 360.0);
  x = c * (1.0 - om_fabs(om_fmod(h / 60.0, 2.0) - 1.0));
  m = v - c;
  if (h >= 0.0 && h < 60.0) { rp = c; gp = x; bp = 0.0; }
  else if (h >= 60.0 && h < 120.0) { rp = x; gp = c; bp = 0.0; }
  else if (h >= 120.0 && h < 180.0) { rp = 0.0; gp = c; bp = x; }
  else if (h >= 180.0 && h < 240.0) { rp = 0.0; gp = x; bp = c; }
  else if (h >= 240.0 && h < 300.0) { rp = x; gp = 0.0; bp = c; }
  else { rp = c; gp = 0.0; bp = x; } /* 300..360; anything else is unreachable!() in the reference (NaN input) */
  rgb[0] = as_u8((rp + m) * 255.0);
  rgb[1] = as_u8((gp + m) * 255.0);
  rgb[2] = as_u8((bp + m) * 255.0);
}

/* SimpleColors::color_for_pixel, coloring/simple.rs:22-45 */
static void simple_color(const atmrt_coloring_t* c, double distance, double elevation, uint8_t rgb[3]) {
  double dist_ratio = distance / c->max_distance;
  if (elevation <= c->water_level) {
    double mul = 1.0 - dist_ratio * 0.6;
    rgb[0] = 0;
    rgb[1] = as_u8(128.0 * mul);
    rgb[2] = as_u8(255.0 * mul);
  } else {
    double elev_ratio = elevation / 4500.0;
    double h = 120.0 - 240.0 * (elev_ratio < 0.0 ? -om_pow(-elev_ratio, 0.65) : om_pow(elev_ratio, 0.65));
    double v = (elev_ratio > 0.7 ? 2.1 - elev_ratio * 2.0 : 0.9 - elev_ratio / 0.7 * 0.2) * (1.0 - dist_ratio * 0.6);
    double s = 1.0 - dist_ratio * 0.9;
    hsv(h, s, v, rgb);
  }
}

/* ColorPalette::{sky_color, water_color, elev_to_color}, coloring/shading.rs:16-83 */
static void palette_sky(int palette, double out[3]) {
  if (palette == ATMRT_PALETTE_LEGACY) { out[0] = 0.11; out[1] = 0.11; out[2] = 0.11; }
  else { out[0] = 0.23; out[1] = 0.41; out[2] = 0.55; }
}
static void palette_water(int palette, double out[3]) {
  if (palette == ATMRT_PALETTE_LEGACY) { out[0] = 0.0; out[1] = 0.5; out[2] = 1.0; }
  else { out[0] = 0.23; out[1] = 0.41; out[2] = 0.55; }
}
static void mix(const double a[3], const double b[3], double prop, double out[3]) { /* a * prop + b * (1.0 - prop) */
  int i;
  for (i = 0; i < 3; i++) out[i] = a[i] * prop + b[i] * (1.0 - prop);
}
static void palette_elev(int palette, double elev, double out[3]) {
  static const double l_green[3] = {0.0, 1.0, 0.0}, l_gy[3] = {0.6, 1.0, 0.0}, l_grey[3] = {0.5, 0.5, 0.5}, l_white[3] = {1.0, 1.0, 1.0};
  static const double i_green[3] = {0.4, 0.8, 0.3}, i_base[3] = {0.77, 0.84, 0.4}, i_mid[3] = {0.41, 0.52, 0.4}, i_top[3] = {0.85, 0.92, 0.95};
  const double *c0, *c1, *c2, *c3;
  double t1 = 300.0, t2, t3 = 1800.0, t4 = 3000.0;
  if (palette == ATMRT_PALETTE_LEGACY) { c0 = l_green; c1 = l_gy; c2 = l_grey; c3 = l_white; t2 = 1200.0; }
  else { c0 = i_green; c1 = i_base; c2 = i_mid; c3 = i_top; t2 = 1000.0; }
  if (elev < t1) memcpy(out, c0, 3 * sizeof(double));
  else if (elev < t2) mix(c1, c0, (elev - t1) / (t2 - t1), out);
  else if (elev < t3) mix(c2, c1, (elev - t2) / (t3 - t2), out);
  else if (elev < t4) mix(c3, c2, (elev - t3) / (t4 - t3), out);
  else memcpy(out, c3, 3 * sizeof(double));
}

/* Shading::color_for_pixel, coloring/shading.rs:118-137 with calc_brightness :111-115 */
static void shading_color(const atmrt_coloring_t* c, const double normal[3], double elevation, uint32_t tag, const double rgba[4],
                          uint8_t rgb[3]) {
  double light_dot = c->light_dir[0] * normal[0] + c->light_dir[1] * normal[1] + c->light_dir[2] * normal[2];
  double brightness, col[3];
  int i;
  light_dot = light_dot >= 0.0 ? light_dot : 0.0;
  brightness = c->ambient_light + (1.0 - c->ambient_light) * light_dot * light_dot;
  if (tag == ATMRT_COLOR_RGBA) { col[0] = rgba[0]; col[1] = rgba[1]; col[2] = rgba[2]; }
  else if (elevation <= c->water_level) palette_water(c->palette, col);
  else palette_elev(c->palette, elevation, col);
  for (i = 0; i < 3; i++) rgb[i] = as_u8(col[i] * brightness * 255.0);
}

static void sky_color(const atmrt_coloring_t* c, uint8_t rgb[3]) {
  if (c->kind == ATMRT_COLORING_SIMPLE) { rgb[0] = rgb[1] = rgb[2] = 28; return; } /* simple.rs:47-49 */
  {
    double s[3];
    int i;
    palette_sky(c->palette, s);
    for (i = 0; i < 3; i++) rgb[i] = as_u8(s[i] * 255.0); /* shading.rs:139-147 */
  }
}

/* fog, renderer/mod.rs:367-376 */
static void fog(double fog_dist, double pixel_dist, uint8_t color[3]) {
  double fog_coeff = 1.0 - om_exp(-pixel_dist / fog_dist);
  int i;
  for (i = 0; i < 3; i++) color[i] = as_u8((double)color[i] * (1.0 - fog_coeff) + 160.0 * fog_coeff);
}

/* add, renderer/mod.rs:378-383 with rgb_to_vec3 / vec3_to_rgb, utils/mod.rs:16-29 */
static void add(uint8_t acc[3], const uint8_t c2[3], double a) {
  int i;
  for (i = 0; i < 3; i++) acc[i] = as_u8(((double)acc[i] / 255.0 + (double)c2[i] / 255.0 * a) * 255.0);
}

/* ConfColoring::into_coloring, params.rs:231-277 */
int oracle_coloring_from_conf(const atmrt_params_t* p, int32_t kind, double water_level, double ambient_light, double light_zenith_angle,
                              double light_dir, int32_t palette, int32_t has_fog, double fog_distance, atmrt_coloring_t* out) {
  memset(out, 0, sizeof *out);
  out->kind = kind;
  out->palette = palette;
  out->water_level = water_level;
  out->max_distance = p->frame.max_distance;
  out->ambient_light = ambient_light;
  out->has_fog = has_fog;
  out->fog_distance = fog_distance;
  if (kind == ATMRT_COLORING_SHADING) {
    double lza = om_to_radians(light_zenith_angle), ld = om_to_radians(light_dir);
    double fa = om_to_radians(p->frame.direction);
    ovec3 n, e, u;
    double front[3], right[3], v[3], len, nn[3], ee[3], uu[3];
    int i;
    oracle_world_directions(&p->earth, p->position.latitude, p->position.longitude, &n, &e, &u);
    nn[0] = n.x; nn[1] = n.y; nn[2] = n.z; ee[0] = e.x; ee[1] = e.y; ee[2] = e.z; uu[0] = u.x; uu[1] = u.y; uu[2] = u.z;
    for (i = 0; i < 3; i++) {
      front[i] = nn[i] * om_cos(fa) + ee[i] * om_sin(fa);
      right[i] = ee[i] * om_cos(fa) - nn[i] * om_sin(fa);
    }
    for (i = 0; i < 3; i++)
      v[i] = -front[i] * om_sin(lza) * om_cos(ld) + right[i] * om_sin(lza) * om_sin(ld) + uu[i] * om_cos(lza);
    len = om_sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    for (i = 0; i < 3; i++) out->light_dir[i] = v[i] / len;
  }
  return 0;
}

/* draw_image, renderer/mod.rs:385-414: front-to-back alpha compositing of every pixel's trace points */
int oracle_draw_image(const atmrt_result_t* r, const atmrt_coloring_t* c, uint8_t* rgb) {
  uint8_t def_color[3];
  size_t p;
  if (c->has_fog) def_color[0] = def_color[1] = def_color[2] = 160; /* fog_color() of both methods */
  else sky_color(c, def_color);
  for (p = 0; p < r->n_pixels; p++) {
    uint8_t result[3] = {0, 0, 0};
    double accum_neg_alpha = 1.0;
    uint64_t k;
    for (k = r->hit_offset[p]; k < r->hit_offset[p] + r->hit_count[p]; k++) {
      uint8_t color[3];
      double alpha = r->rgba[4 * k + 3]; /* PixelColor::alpha, generators/mod.rs:52-57 */
      if (c->kind == ATMRT_COLORING_SIMPLE) simple_color(c, r->distance[k], r->elevation[k], color);
      else shading_color(c, &r->normal[3 * k], r->elevation[k], r->color_tag[k], &r->rgba[4 * k], color);
      if (c->has_fog) fog(c->fog_distance, r->path_length[k], color);
      add(result, color, accum_neg_alpha * alpha);
      accum_neg_alpha *= 1.0 - alpha;
    }
    add(result, def_color, accum_neg_alpha);
    memcpy(&rgb[3 * p], result, 3);
  }
  return 0;
}

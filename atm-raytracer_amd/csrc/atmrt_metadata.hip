// atmrt_metadata.hip — SURVEY §8(f) rank 2: the `result` half of the reference's metadata file.
//
// generator::output_metadata (src/generator/mod.rs:26-45) writes gzip(bincode::serialize(&AllData{params, result})), and
// `view` reads it back (src/viewer/mod.rs:17-29).  bincode 1.x with its default options is a fixed layout: integers
// little-endian at their own width, f64 as 8 LE bytes, a sequence as a u64 length followed by its elements, a struct as its
// fields in declaration order without framing, an enum as a u32 variant index followed by the variant's fields.  With the
// derives of generators/mod.rs:13-49 and object/mod.rs:133-140 that gives, for result: Vec<Vec<ResultPixel>>:
//
//   u64 H, then per row: u64 W, then per pixel:
//     f64 elevation_angle, f64 azimuth, u64 n, then per trace point:
//       f64 lat, lon, distance, elevation, path_length
//       normal: Vector3<f64>            [u64 3] f64 x, y, z      <- the length prefix is nalgebra's (crate absent: UNPINNED;
//                                                                   `vector3_len_prefix` selects either form)
//       color: PixelColor               u32 0, f64 alpha                 (Terrain(alpha))
//                                       u32 1, f64 r, g, b, a            (Rgba(Color))
//
// Host code only (no device work): encoding 8 M pixels from the SoA result is a bandwidth-bound loop.  The `params` half and
// the gzip framing are written by the host harness (atm-raytracer_amd/metadata.py); see DESIGN.md for the bytes of it that
// remain unpinned.
#include <cstdint>
#include <cstring>

#include "../../include/atmrt.h"
#include "atmrt_hostmem.h"

namespace {

struct Writer {
  uint8_t* p;
  void u32(uint32_t v) { memcpy(p, &v, 4); p += 4; }
  void u64(uint64_t v) { memcpy(p, &v, 8); p += 8; }
  void f64(double v) { memcpy(p, &v, 8); p += 8; }
};

struct Reader {
  const uint8_t* p;
  const uint8_t* end;
  bool ok = true;
  bool need(size_t n) {
    if ((size_t)(end - p) < n) ok = false;
    return ok;
  }
  uint32_t u32() { uint32_t v = 0; if (need(4)) { memcpy(&v, p, 4); p += 4; } return v; }
  uint64_t u64() { uint64_t v = 0; if (need(8)) { memcpy(&v, p, 8); p += 8; } return v; }
  double f64() { double v = 0; if (need(8)) { memcpy(&v, p, 8); p += 8; } return v; }
};

size_t tp_bytes(uint32_t tag, bool prefix) { return 40 + (prefix ? 8 : 0) + 24 + 4 + (tag == ATMRT_COLOR_TERRAIN ? 8 : 32); }

} // namespace

extern "C" int atmrt_result_encode_bincode(const atmrt_result_t* r, int32_t vector3_len_prefix, uint8_t* dst, size_t capacity,
                                           size_t* n_bytes) {
  if (!r || !n_bytes) return ATMRT_ERR_INVALID_ARGUMENT;
  const size_t npx = (size_t)r->width * r->height;
  if (npx != r->n_pixels || (npx && (!r->azimuth || !r->elevation_angle || !r->hit_count || !r->hit_offset)))
    return ATMRT_ERR_INVALID_ARGUMENT;
  const bool prefix = vector3_len_prefix != 0;
  size_t total = 8 + (size_t)r->height * 8 + npx * 24;
  for (size_t p = 0; p < npx; p++) {
    const uint64_t k0 = r->hit_offset[p], n = r->hit_count[p];
    if (k0 + n > r->n_hits) return ATMRT_ERR_INVALID_ARGUMENT;
    for (uint64_t k = k0; k < k0 + n; k++) {
      if (r->color_tag[k] > ATMRT_COLOR_RGBA) return ATMRT_ERR_INVALID_ARGUMENT;
      total += tp_bytes(r->color_tag[k], prefix);
    }
  }
  *n_bytes = total;
  if (!dst) return ATMRT_OK; // size query
  if (capacity < total) return ATMRT_ERR_INVALID_ARGUMENT;
  Writer w{dst};
  w.u64(r->height);
  for (uint32_t y = 0; y < r->height; y++) {
    w.u64(r->width);
    for (uint32_t x = 0; x < r->width; x++) {
      const size_t p = (size_t)y * r->width + x;
      w.f64(r->elevation_angle[p]);
      w.f64(r->azimuth[p]);
      const uint64_t k0 = r->hit_offset[p], n = r->hit_count[p];
      w.u64(n);
      for (uint64_t k = k0; k < k0 + n; k++) {
        w.f64(r->lat[k]);
        w.f64(r->lon[k]);
        w.f64(r->distance[k]);
        w.f64(r->elevation[k]);
        w.f64(r->path_length[k]);
        if (prefix) w.u64(3);
        w.f64(r->normal[3 * k]);
        w.f64(r->normal[3 * k + 1]);
        w.f64(r->normal[3 * k + 2]);
        w.u32(r->color_tag[k]);
        if (r->color_tag[k] == ATMRT_COLOR_TERRAIN) {
          w.f64(r->rgba[4 * k + 3]);
        } else {
          for (int c = 0; c < 4; c++) w.f64(r->rgba[4 * k + c]);
        }
      }
    }
  }
  return (size_t)(w.p - dst) == total ? ATMRT_OK : ATMRT_ERR_INVALID_ARGUMENT;
}

extern "C" int atmrt_result_decode_bincode(const uint8_t* src, size_t n_bytes, int32_t vector3_len_prefix, atmrt_result_t* out,
                                           size_t* consumed) {
  if (!src || !out) return ATMRT_ERR_INVALID_ARGUMENT;
  memset(out, 0, sizeof *out);
  const bool prefix = vector3_len_prefix != 0;
  uint64_t H = 0, W = 0, n_hits = 0;
  for (int pass = 0; pass < 2; pass++) { // pass 0: validate and count; pass 1: fill
    Reader rd{src, src + n_bytes};
    H = rd.u64();
    if (!rd.ok || H > 65535) return ATMRT_ERR_FORMAT;
    uint64_t k = 0;
    for (uint64_t y = 0; y < H; y++) {
      const uint64_t w = rd.u64();
      if (!rd.ok || w > 65535 || (y > 0 && w != W)) return ATMRT_ERR_FORMAT; // ragged rows cannot be a frame
      W = w;
      for (uint64_t x = 0; x < W; x++) {
        const size_t p = (size_t)(y * W + x);
        const double ea = rd.f64(), az = rd.f64();
        const uint64_t n = rd.u64();
        if (!rd.ok || n > (uint64_t)(rd.end - rd.p) / 76) return ATMRT_ERR_FORMAT;
        if (pass) {
          out->elevation_angle[p] = ea;
          out->azimuth[p] = az;
          out->hit_count[p] = (uint32_t)n;
          out->hit_offset[p] = k;
        }
        for (uint64_t j = 0; j < n; j++, k++) {
          double v[5];
          for (double& q : v) q = rd.f64();
          if (prefix && rd.u64() != 3) return ATMRT_ERR_FORMAT;
          double nrm[3];
          for (double& q : nrm) q = rd.f64();
          const uint32_t tag = rd.u32();
          double rgba[4] = {0.0, 0.0, 0.0, 0.0};
          if (tag == ATMRT_COLOR_TERRAIN) rgba[3] = rd.f64();
          else if (tag == ATMRT_COLOR_RGBA) for (double& q : rgba) q = rd.f64();
          else return ATMRT_ERR_FORMAT;
          if (!rd.ok) return ATMRT_ERR_FORMAT;
          if (pass) {
            out->lat[k] = v[0];
            out->lon[k] = v[1];
            out->distance[k] = v[2];
            out->elevation[k] = v[3];
            out->path_length[k] = v[4];
            memcpy(&out->normal[3 * k], nrm, sizeof nrm);
            out->color_tag[k] = tag;
            memcpy(&out->rgba[4 * k], rgba, sizeof rgba);
          }
        }
      }
    }
    if (!rd.ok) return ATMRT_ERR_FORMAT;
    if (pass == 0) {
      n_hits = k;
      if (H == 0) W = 0;
      if (atmrt_internal_result_alloc(out, (uint32_t)W, (uint32_t)H, n_hits)) return ATMRT_ERR_INVALID_ARGUMENT;
    } else if (consumed) {
      *consumed = (size_t)(rd.p - src);
    }
  }
  return ATMRT_OK;
}

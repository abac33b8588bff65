// atmrt_ctx.h — internal: the context behind the C ABI (include/atmrt.h), shared by the translation units that implement it:
// atmrt_api.hip (one device: terrain store, frame set-up, launch sequences) and atmrt_multi.hip (several devices: column tiles,
// the RCCL all-gather, image assembly).
#pragma once

#include <cstdarg>
#include <cstdio>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "atmrt_kernels.h"

namespace atmrt {

struct HostTile {
  int n_lat = 0, n_lon = 0;
  std::vector<int16_t> posts; // [n_lat][n_lon]
};
// Terrain (terrain/mod.rs:55-57): tiles keyed by integer degrees.  One store may serve several contexts (the sub-contexts of a
// multi-device context upload the same mosaic to their own HBM): `generation` tells a context that its copy is stale.
struct TileStore {
  std::map<std::pair<int, int>, HostTile> tiles;
  uint64_t generation = 1;
};

// grow-only device buffer
struct DevBuf {
  void* ptr = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
    size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipMalloc(&ptr, want);
    if (e == hipSuccess) cap = want;
    return e;
  }
  void release() {
    if (ptr) (void)hipFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const { return static_cast<T*>(ptr); }
  DevBuf() = default;
  DevBuf(const DevBuf&) = delete;
  DevBuf& operator=(const DevBuf&) = delete;
  DevBuf(DevBuf&& o) noexcept : ptr(o.ptr), cap(o.cap) { o.ptr = nullptr, o.cap = 0; }
  DevBuf& operator=(DevBuf&& o) noexcept { // std::swap of two buffers (run_interpolating keeps the lattice result that way)
    if (this != &o) {
      release();
      ptr = o.ptr, cap = o.cap;
      o.ptr = nullptr, o.cap = 0;
    }
    return *this;
  }
  ~DevBuf() { release(); } // atmrt_ctx_destroy makes the context's device current before the context (and its buffers) goes
};

// An owning copy of an atmrt_atmosphere_t (the ABI struct borrows its function table and spline points from the caller).
struct AtmDef {
  atmrt_atmosphere_t pod{};
  std::vector<atmrt_temp_function_t> functions;
  std::vector<std::vector<double>> xs, ys;
  void assign(const atmrt_atmosphere_t& a) {
    pod = a;
    functions.assign(a.functions, a.functions + a.n_functions);
    xs.assign(functions.size(), {});
    ys.assign(functions.size(), {});
    for (size_t j = 0; j < functions.size(); j++) {
      atmrt_temp_function_t& fn = functions[j];
      if (fn.kind == ATMRT_TEMP_SPLINE && fn.n_points > 0 && fn.point_altitude && fn.point_temperature) {
        xs[j].assign(fn.point_altitude, fn.point_altitude + fn.n_points);
        ys[j].assign(fn.point_temperature, fn.point_temperature + fn.n_points);
        fn.point_altitude = xs[j].data();
        fn.point_temperature = ys[j].data();
      } else {
        fn.point_altitude = fn.point_temperature = nullptr;
      }
    }
    pod.functions = functions.data();
  }
  AtmDef() = default;
  AtmDef(const AtmDef&) = delete;
  AtmDef& operator=(const AtmDef&) = delete;
};

struct MultiGroup; // atmrt_multi.hip: the devices of a multi-device context and their worker threads
struct Comm;       // atmrt_multi.hip: this context's place among the ranks that share one frame

} // namespace atmrt

struct atmrt_ctx {
  int device = 0;
  hipStream_t stream = nullptr, stream2 = nullptr;
  hipEvent_t ev_fork = nullptr, ev_join = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
  hipEvent_t ev[10] = {};
  hipEvent_t ev_seg[atmrt::FAST_SEGMENTS] = {}; // a path segment is integrated (stream2) -> its intersect scan may start
  hipEvent_t ev_scan[2 * atmrt::FAST_SEGMENTS] = {}; // begin / end of every scan segment (after its wait), for intersect_ms
  int scan_segments = 0;                       // segments of the last pipelined frame (0: ev[4]..ev[5] time the scan)
  atmrt_timings_t timings{};
  atmrt_frame_stats_t stats{};
  bool inject_failure = false; // atmrt_debug_fail_next_frame
  std::string error;

  std::shared_ptr<atmrt::TileStore> terrain = std::make_shared<atmrt::TileStore>();
  uint64_t terrain_uploaded = 0; // generation of the mosaic in d_posts (0: none)
  atmrt::DevBuf d_posts, d_tiles, d_cells;
  atmrt::TerrainView tv{};

  bool have_params = false;
  atmrt_params_t params{};       // as the caller set them (a rank of a shared frame: the WHOLE image; its columns are in `comm`)
  atmrt::AtmDef atm_def;
  std::vector<uint8_t> atm_def_bytes; // the definition as it was last set, byte for byte (atmrt_set_atmosphere)
  uint64_t atm_def_serial = 0;        // counts its changes
  atmrt::AtmTableBuf atm;
  struct AtmKey {                     // what the compiled table in `atm` / d_atm was built from (prepare_frame)
    uint64_t def_serial;
    double wavelength, step, radius;
    int32_t spherical;
    int32_t _pad = 0;
  } atm_key{};
  bool atm_key_valid = false;
  atmrt::Earth earth{};
  atmrt::Pinhole pinhole{};
  std::vector<double> xs;
  int n_t = 0, n_path_cap = 0;
  bool xs_dirty = true;

  // last generated frame (for atmrt_draw_image)
  bool last_valid = false, last_packed = false;
  size_t last_npx = 0;
  int last_wl = 0, last_h = 0, last_c0 = 0;
  double last_alpha = 1.0;
  atmrt::DensePlanes last_dense{};
  atmrt::PackedHits last_hits{};
  const uint64_t* last_offset = nullptr;
  uint64_t last_nhits = 0;

  std::vector<atmrt::ObjectDev> objects; // host image of the device table (altitude kind in _pad until k_resolve)
  std::vector<uint8_t> textures;         // RGBA8 pool
  bool objects_dirty = true;

  // several devices / ranks (atmrt_multi.hip); both null for a plain one-device context
  atmrt::MultiGroup* multi = nullptr; // this is the PARENT of a multi-device context: every entry point forwards to its children
  atmrt::Comm* comm = nullptr;        // this context computes one column tile of a frame shared with other ranks

  // workspace
  atmrt::DevBuf d_xs, d_alt, d_colcalc, d_prof, d_pelev, d_plen, d_npath, d_hit_step, d_hit_offset, d_scan_tmp, d_counters,
      d_list_step, d_list_pixel, d_rect_rec, d_dense, d_packed, d_io, d_objects, d_textures, d_plat, d_plon,
      d_ccount, d_coffset, d_clist, d_px_steps, d_atm, d_interp, d_lat_dense, d_lat_packed, d_lat_offset, d_slot_step, d_slot_rec,
      d_overflow, d_slot_pixel, d_slot_packed, d_pelev_t, d_plen_t, d_col_cand, d_col_ncand, d_path_seg, d_dprev, d_step_prop,
      d_blend_arena, d_object_rays, d_col_lo, d_col_hi, d_traced, d_slice, d_overflow_arena, d_step_ctx;

  int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    error = buf;
    return code;
  }
};

#define HIP_TRY(ctx, expr)                                                                               \
  do {                                                                                                   \
    hipError_t e_ = (expr);                                                                              \
    if (e_ != hipSuccess) return (ctx)->fail(ATMRT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
  } while (0)


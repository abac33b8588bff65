// atmrt_kernels.hip — gfx950 (MI355X, CDNA4) kernels of the ray-marching path.
//
// Wavefront = 64 lanes everywhere.  All arithmetic is IEEE binary64 with contraction disabled, in
// the operation order of atmrt_core.h, so results are bit-identical to the CPU checker.
//
// Fast generator (fast.rs:22-98) as four kernels:
//   k_fast_columns    per-column DirectionalCalc                  (utils.rs:183-189)
//   k_terrain_profile phase A: terrain elevation per (sample, column)   (utils.rs:176-199, 84)
//   k_fast_paths      phase B: one RK4 ray per image row          (utils.rs:136-174)
//   k_fast_intersect  phase C: sign-change scan of get_single_pixel     (utils.rs:211-240)
//   k_fast_finalize   TracePoint at the bracketing samples only   (utils.rs:108-125, 15-40)
// Rectilinear generator (rectilinear.rs:102-186): k_rect_march, one ray per lane.
// The per-object collision code is called out of line in this translation unit as well (only k_fast_trace uses it): inlined it
// takes the Fast tracer to 256 VGPRs / 2 waves per SIMD; out of line at 3 waves config 5 runs 26.0 -> 23.4 ms (Fast) and
// 73.4 -> 67.7 ms (InterpolatingRectilinear); 4 waves 26.9 ms.
#define ATMRT_OBJ_FN __attribute__((noinline))
#include "atmrt_device.h"
#include "atmrt_render.h"

namespace atmrt {

// ---------------------------------------------------------------------------------------------
// set-up: Altitude::abs for the observer and every object (params.rs:23-30, object/mod.rs:166-175)
// ---------------------------------------------------------------------------------------------
__global__ void k_resolve(Frame f, double* alt, ObjectDev* objects) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0) {
    const atmrt_position_t& pos = f.p.position;
    *alt = pos.altitude_kind == ATMRT_ALT_ABSOLUTE
               ? pos.altitude
               : terrain_elev_or_zero(f.tv, pos.latitude, pos.longitude) + pos.altitude;
  }
  if (i < f.n_objects) { // the host stores the altitude kind in _pad and the configured altitude in elev
    ObjectDev o = objects[i];
    if (o._pad == ATMRT_ALT_RELATIVE) o.elev = terrain_elev_or_zero(f.tv, o.lat, o.lon) + o.elev;
    o._pad = ATMRT_ALT_ABSOLUTE;
    object_derive(f.earth, f.p.simulation_step, o);
    objects[i] = o;
  }
}

// ---------------------------------------------------------------------------------------------
// Fast generator
// ---------------------------------------------------------------------------------------------
__global__ void k_fast_columns(Frame f, DirCalc* colcalc) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= f.wl) return;
  double dir = frame_col_dir(f, x);
  DirCalc c;
  dircalc_new(f.earth, f.p.position.latitude, f.p.position.longitude, dir, c);
  colcalc[x] = c;
}

// Phase A.  A wavefront covers 64 adjacent columns at one sample index, so its profile store is one
// coalesced 512-byte row segment and its terrain gathers fall on neighbouring posts of the mosaic.
constexpr int PROFILE_SAMPLES_PER_BLOCK = 16;
template <int CALC, bool STORE_LL>
__global__ __launch_bounds__(256) void k_terrain_profile(Frame f, const DirCalc* __restrict__ colcalc,
                                                         double* __restrict__ prof, double* __restrict__ plat,
                                                         double* __restrict__ plon) {
  // sample blocks on grid.x (up to 4e6 / 16 of them), column tiles on grid.y (at most 1024): grid.y is limited to 65535
  int x = blockIdx.y * 64 + (threadIdx.x & 63);
  int sub = threadIdx.x >> 6;
  if (x >= f.wl) return;
  const Earth e = earth_for<CALC>(f);
  const DirCalc c = colcalc[x];
  int i0 = blockIdx.x * PROFILE_SAMPLES_PER_BLOCK;
  for (int k = sub; k < PROFILE_SAMPLES_PER_BLOCK; k += 4) {
    int i = i0 + k;
    if (i >= f.n_t) break;
    double lat, lon;
    coords_at_dist(e, c, f.xs[i], lat, lon);
    prof[(size_t)i * f.wl + x] = terrain_elev_or_zero(f.tv, lat, lon);
    if (STORE_LL) {
      plat[(size_t)i * f.wl + x] = lat;
      plon[(size_t)i * f.wl + x] = lon;
    }
  }
}

// Phase B (k_fast_paths, one RK4 ray per image row on an octet of lanes) lives in atmrt_paths.hip.

// Phase C.  Lanes = 64 adjacent columns, each wavefront owns RR adjacent rows.  The terrain value prof[i][x] is
// loaded once per lane (coalesced) and reused for RR rows; the ray elevations pelev[y][i..i+CH) are wave-uniform and
// arrive as wide scalar loads.  The body is branch-free (selects), samples are processed CH at a time and the
// all-lanes-finished test (MODE 0) runs once per chunk.  MODE 0: opaque terrain, first sign change only.
// MODE 1: count every sign change (terrain_alpha < 1).
template <int RR, int MODE>
__global__ __launch_bounds__(256) void k_fast_intersect(Frame f, const double* __restrict__ prof,
                                                        const double* __restrict__ pelev,
                                                        const int32_t* __restrict__ npath,
                                                        int32_t* __restrict__ hit_step,
                                                        uint32_t* __restrict__ hit_count,
                                                        uint32_t* __restrict__ px_steps,
                                                        unsigned long long* __restrict__ counters, double* __restrict__ dprev_state,
                                                        int i_begin, int i_end, int last_segment, uint32_t* __restrict__ slot_step,
                                                        uint32_t* __restrict__ slot_tag, const uint8_t* __restrict__ traced) {
  // slot_tag != nullptr (scenes with objects, one launch over all samples): the slots are the general tracer's arena — entry
  // p * RECT_SLOTS + j, tag ATMRT_COLOR_TERRAIN — and the pixels marked in `traced` are k_fast_trace's: their ray-steps are not
  // counted here (whatever this scan stores for them is overwritten).
  // Samples i_begin .. i_end - 1 (the frame is scanned in the segments in which its ray paths are integrated, so that the scan
  // of one segment overlaps the integration of the next); between segments a pixel's state is the difference at its last
  // sample (dprev_state), its first hit (hit_step) and its count (hit_count).
  constexpr int CH = 4;
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  // Tile order: blockIdx.x = column tile, blockIdx.y = row group.  Consecutive workgroups share a row group (its ray
  // elevations stay in the scalar cache / L2) and sweep the column tiles of the terrain profile.  An explicitly XCD-aware
  // order (XCD r owns column tiles r, r+8, ... and walks all their row groups) was measured and rejected: FETCH_SIZE 0.50 GB
  // vs 0.44 GB per frame and 2.46 ms vs 2.21 ms — the whole working set (65 MB profile + 33 MB paths) sits in the 256 MB
  // Infinity Cache and the kernel is not bandwidth-limited.
  const int x = blockIdx.x * 64 + lane;
  const bool xok = x < f.wl;
  const int xc = xok ? x : f.wl - 1;
  const int y0 = (blockIdx.y * 4 + wave) * RR;
  const int cap = f.n_path_cap;
  const size_t wl = (size_t)f.wl;
  const size_t plane_px = wl * (size_t)f.h;
  if (y0 >= f.h) return;

  // rows past the image repeat the last row (identical work, never stored), so the fast loop stays uniform
  const double* prow[RR];
  int nrow[RR];
  double dprev[RR];
  int first[RR];
  unsigned cnt[RR];
  int nmin = 0x7fffffff, nmax = 0;
  const double t0 = prof[xc];
#pragma unroll
  for (int r = 0; r < RR; r++) {
    int y = y0 + r < f.h ? y0 + r : f.h - 1;
    int n = npath[y];
    n = n < f.n_t ? n : f.n_t; // Iterator::zip, fast.rs:59-62
    nrow[r] = n;
    nmin = n < nmin ? n : nmin;
    nmax = n > nmax ? n : nmax;
    prow[r] = pelev + (size_t)y * cap;
    first[r] = -1;
    cnt[r] = 0;
    dprev[r] = prow[r][0] - t0;
  }
  int nfound = 0; // rows of this lane that have their first hit (MODE 0)
  if (i_begin > 1) { // resume from the previous segment
#pragma unroll
    for (int r = 0; r < RR; r++) {
      const size_t p = (size_t)(y0 + r < f.h ? y0 + r : f.h - 1) * wl + xc;
      dprev[r] = dprev_state[p];
      first[r] = hit_step[p];
      if (MODE == 0) nfound += first[r] >= 0 ? 1 : 0;
      else cnt[r] = hit_count[p];
    }
  }
  nmin = nmin < i_end ? nmin : i_end;
  nmax = nmax < i_end ? nmax : i_end;
  bool alldone = MODE == 0 && __all(nfound == RR);
  int i = i_begin;
  for (; !alldone && i + CH <= nmin; i += CH) {
    double t[CH];
#pragma unroll
    for (int k = 0; k < CH; k++) t[k] = prof[(size_t)(i + k) * wl + xc];
#pragma unroll
    for (int r = 0; r < RR; r++) {
      const double* pr = prow[r] + i;
      double p[CH];
#pragma unroll
      for (int k = 0; k < CH; k++) p[k] = pr[k];
      unsigned hitbits = 0;
      const unsigned cnt0 = cnt[r];
#pragma unroll
      for (int k = 0; k < CH; k++) {
        const double d = p[k] - t[k];
        const bool hit = dprev[r] * d < 0.0; // utils.rs:222
        const bool nh = hit && first[r] < 0;
        first[r] = nh ? i + k - 1 : first[r];
        if (MODE == 0) nfound += nh ? 1 : 0;
        else {
          cnt[r] += hit ? 1u : 0u;
          hitbits |= hit ? 1u << k : 0u;
        }
        dprev[r] = d;
      }
      if (MODE != 0 && hitbits && xok && y0 + r < f.h) { // rare: keep the first RECT_SLOTS crossings of the pixel for the list
        unsigned c = cnt0;
        const size_t pp = (size_t)(y0 + r) * wl + x;
        for (unsigned hb = hitbits; hb && c < (unsigned)RECT_SLOTS; hb &= hb - 1, c++) {
          const size_t q = slot_tag ? pp * RECT_SLOTS + c : (size_t)c * plane_px + pp;
          slot_step[q] = (uint32_t)(i + __builtin_ctz(hb) - 1);
          if (slot_tag) slot_tag[q] = ATMRT_COLOR_TERRAIN;
        }
      }
    }
    if (MODE == 0 && __all(nfound == RR)) {
      alldone = true;
      break;
    }
  }
  if (!alldone) { // remainder of the chunking and rows whose path ended early (ray below -1000 m, utils.rs:167)
    for (; i < nmax; i++) {
      const double t = prof[(size_t)i * wl + xc];
#pragma unroll
      for (int r = 0; r < RR; r++) {
        if (i < nrow[r]) {
          const double d = prow[r][i] - t;
          const bool hit = dprev[r] * d < 0.0;
          const bool nh = hit && first[r] < 0;
          first[r] = nh ? i - 1 : first[r];
          if (MODE != 0 && hit) {
            if (cnt[r] < (unsigned)RECT_SLOTS && xok && y0 + r < f.h) {
              const size_t pp = (size_t)(y0 + r) * wl + x;
              const size_t q = slot_tag ? pp * RECT_SLOTS + cnt[r] : (size_t)cnt[r] * plane_px + pp;
              slot_step[q] = (uint32_t)(i - 1);
              if (slot_tag) slot_tag[q] = ATMRT_COLOR_TERRAIN;
            }
            cnt[r]++;
          }
          dprev[r] = d;
        }
      }
    }
  }

  if (!last_segment) {
#pragma unroll
    for (int r = 0; r < RR; r++) {
      int y = y0 + r;
      if (y < f.h && xok) {
        size_t p = (size_t)y * f.wl + x;
        dprev_state[p] = dprev[r];
        hit_step[p] = first[r];
        if (MODE != 0) hit_count[p] = cnt[r];
      }
    }
    return;
  }
  unsigned long long steps = 0;
#pragma unroll
  for (int r = 0; r < RR; r++) {
    int y = y0 + r;
    if (y < f.h && xok) {
      size_t p = (size_t)y * f.wl + x;
      hit_step[p] = first[r];
      unsigned st;
      if (MODE == 0) {
        hit_count[p] = first[r] >= 0 ? 1u : 0u;
        st = first[r] >= 0 ? (unsigned)(first[r] + 1) : (unsigned)(nrow[r] > 0 ? nrow[r] - 1 : 0);
        if (slot_tag && first[r] >= 0) { // opaque terrain in a scene with objects: the first crossing is the pixel's only slot
          slot_step[p * RECT_SLOTS] = (uint32_t)first[r];
          slot_tag[p * RECT_SLOTS] = ATMRT_COLOR_TERRAIN;
        }
      } else {
        hit_count[p] = cnt[r];
        st = (unsigned)(nrow[r] > 0 ? nrow[r] - 1 : 0);
      }
      if (traced && traced[p]) st = 0;
      steps += st;
      if (px_steps) px_steps[p] = st;
    }
  }
  steps = wave_sum(steps); // one atomic per wavefront: 16 K of them at the headline size, spread over the kernel's 2 ms
  if (lane == 0 && steps) atomicAdd(&counters[0], steps);
}

// The two samples that bracket step s of pixel (x, y), from the caches
static __device__ __forceinline__ TracePointDev fast_hit(const Frame& f, const Earth& e, const DirCalc& c, const double* prof, const double* pelev,
                                         const double* plen, int x, int y, int s) {
  double lat0, lon0, lat1, lon1;
  double d0 = f.xs[s], d1 = f.xs[s + 1];
  coords_at_dist(e, c, d0, lat0, lon0);
  coords_at_dist(e, c, d1, lat1, lon1);
  double te0 = prof[(size_t)s * f.wl + x], te1 = prof[(size_t)(s + 1) * f.wl + x];
  size_t base = (size_t)y * f.n_path_cap;
  double re0 = pelev[base + s], re1 = pelev[base + s + 1];
  // TracingState::new(&first_terrain, first_path.elev, 0.0, 0.0) (utils.rs:208): path[0] is (0, alt, 0) anyway
  double pl0 = plen[base + s], pl1 = plen[base + s + 1];
  return terrain_trace_point(f, e, lat0, lon0, te0, re0, d0, pl0, lat1, lon1, te1, re1, d1, pl1);
}

template <int CALC>
__global__ __launch_bounds__(256) void k_fast_finalize(Frame f, const DirCalc* __restrict__ colcalc,
                                                       const double* __restrict__ prof,
                                                       const double* __restrict__ pelev,
                                                       const double* __restrict__ plen,
                                                       const int32_t* __restrict__ hit_step, DensePlanes out) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y;
  if (x >= f.wl) return;
  size_t plane = (size_t)f.wl * f.h;
  size_t p = (size_t)y * f.wl + x;
  out.azimuth[p] = frame_azimuth(f, x);
  out.elevation_angle[p] = frame_row_elev(f, y);
  int s = hit_step[p];
  if (s < 0) {
    store_dense_miss(out, p, plane);
    return;
  }
  const DirCalc c = colcalc[x];
  store_dense(out, p, plane, fast_hit(f, earth_for<CALC>(f), c, prof, pelev, plen, x, y, s));
}

// terrain_alpha < 1: the list of every sign change of every pixel (step index + pixel).  The counting scan kept the first
// RECT_SLOTS crossings of each pixel; k_fast_gather_steps moves those, and only pixels with more are scanned a second time.
__global__ __launch_bounds__(256) void k_fast_gather_steps(Frame f, const uint32_t* __restrict__ hit_count,
                                                           const uint64_t* __restrict__ hit_offset,
                                                           const uint32_t* __restrict__ slot_step, uint32_t* __restrict__ list_step,
                                                           uint32_t* __restrict__ list_pixel) {
  const size_t plane = (size_t)f.wl * f.h;
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const uint32_t n = hit_count[p];
  if (n > (uint32_t)RECT_SLOTS) return;
  const uint64_t k = hit_offset[p];
  for (uint32_t j = 0; j < n; j++) {
    list_step[k + j] = slot_step[(size_t)j * plane + p];
    list_pixel[k + j] = (uint32_t)p;
  }
}

__global__ __launch_bounds__(256) void k_fast_list(Frame f, const double* __restrict__ prof,
                                                   const double* __restrict__ pelev,
                                                   const int32_t* __restrict__ npath,
                                                   const uint64_t* __restrict__ hit_offset,
                                                   const uint32_t* __restrict__ hit_count,
                                                   uint32_t* __restrict__ list_step,
                                                   uint32_t* __restrict__ list_pixel) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y;
  if (x >= f.wl) return;
  int n = npath[y];
  n = n < f.n_t ? n : f.n_t;
  size_t p = (size_t)y * f.wl + x;
  if (hit_count[p] <= (uint32_t)RECT_SLOTS) return; // listed from the slots of the counting scan (k_fast_gather_steps)
  uint64_t k = hit_offset[p];
  const double* row = pelev + (size_t)y * f.n_path_cap;
  double dprev = row[0] - prof[x];
  for (int i = 1; i < n; i++) {
    double d = row[i] - prof[(size_t)i * f.wl + x];
    if (dprev * d < 0.0) {
      list_step[k] = (uint32_t)(i - 1);
      list_pixel[k] = (uint32_t)p;
      k++;
    }
    dprev = d;
  }
}

template <int CALC>
__global__ __launch_bounds__(256) void k_fast_finalize_list(Frame f, uint64_t n_hits,
                                                            const DirCalc* __restrict__ colcalc,
                                                            const double* __restrict__ prof,
                                                            const double* __restrict__ pelev,
                                                            const double* __restrict__ plen,
                                                            const uint32_t* __restrict__ list_step,
                                                            const uint32_t* __restrict__ list_pixel, PackedHits packed) {
  uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_hits) return;
  if (f.n_objects && packed.color_tag[k] != ATMRT_COLOR_TERRAIN) return; // object points are already complete
  uint32_t p = list_pixel[k];
  int x = (int)(p % (uint32_t)f.wl), y = (int)(p / (uint32_t)f.wl);
  const DirCalc c = colcalc[x];
  store_packed(packed, k, fast_hit(f, earth_for<CALC>(f), c, prof, pelev, plen, x, y, (int)list_step[k]),
               f.p.terrain_alpha);
}

// azimuth / elevation planes and the dense first-hit view of a packed multi-hit result
__global__ __launch_bounds__(256) void k_dense_from_packed(Frame f, const uint64_t* __restrict__ hit_offset,
                                                           PackedHits packed, DensePlanes out, int fast_angles) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y;
  if (x >= f.wl) return;
  size_t plane = (size_t)f.wl * f.h;
  size_t p = (size_t)y * f.wl + x;
  if (fast_angles) {
    out.azimuth[p] = frame_azimuth(f, x);
    out.elevation_angle[p] = frame_row_elev(f, y);
  }
  if (out.hit_count[p] == 0) {
    store_dense_miss(out, p, plane);
    return;
  }
  uint64_t k = hit_offset[p];
  TracePointDev tp;
  tp.lat = packed.lat[k];
  tp.lon = packed.lon[k];
  tp.distance = packed.distance[k];
  tp.elevation = packed.elevation[k];
  tp.path_length = packed.path_length[k];
  tp.normal = v3(packed.normal[3 * k], packed.normal[3 * k + 1], packed.normal[3 * k + 2]);
  store_dense(out, p, plane, tp);
}

// Fast, phase A extras: which objects are close to each terrain sample (TerrainData::from_lat_lon, utils.rs:74-80)
// Objects that can be close to ANY sample of a column (ray_candidates: exact superset, from the column's ground track), so that
// the per-sample proximity filter below tests a handful of objects instead of all of them.  ncand[x] < 0: no list for the
// column (more than COL_CAND candidates, or a DirectionalCalc without the pre-filter) — every object is tested.
constexpr int COL_CAND = 64;
template <int CALC>
__global__ __launch_bounds__(64) void k_column_candidates(Frame f, const DirCalc* __restrict__ colcalc, int32_t* __restrict__ ccand,
                                                          int32_t* __restrict__ ncand, double* __restrict__ cand_lo,
                                                          double* __restrict__ cand_hi, unsigned long long* __restrict__ counters) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= f.wl) return;
  int cand[COL_CAND];
  double lo[COL_CAND], hi[COL_CAND]; // distances at which a sample of the column can be close to the candidate (k_fast_flag_rows)
  int n = 0;
  const bool ok = ray_candidates<CALC, COL_CAND>(f, f.earth, colcalc[x], cand, n, lo, hi);
  ncand[x] = ok ? n : -1;
  if (!ok && n == COL_CAND) atomicAdd(&counters[5], 1ull); // statistics only (atmrt_last_stats)
  if (ok)
    for (int q = 0; q < n; q++) {
      ccand[(size_t)x * COL_CAND + q] = cand[q];
      cand_lo[(size_t)x * COL_CAND + q] = lo[q];
      cand_hi[(size_t)x * COL_CAND + q] = hi[q];
    }
}

// Fast generator with scene objects: which pixels can have a step that involves an object?  An object is tested at the sample pair
// (i - 1, i) of column x only if it is close to one of the two samples — possible only while the sample's distance lies in the
// candidate's interval [lo, hi] (candidate_interval: an exact superset) — and only by rows whose ray segment enters the object's height
// band (object_out_of_band, the tracer's own test).  Every other pixel's result is its terrain crossings alone: the plain intersect
// scan's.  One block per column and 256 rows: the column's candidates are wave-uniform, the ray elevations are read sample-major
// (coalesced over rows).  A column without a candidate list (more than COL_CAND candidates, or an earth model without the
// pre-filter) flags all its rows.
__global__ __launch_bounds__(256) void k_fast_flag_rows(Frame f, const int32_t* __restrict__ ccand, const int32_t* __restrict__ ncand,
                                                        const double* __restrict__ cand_lo, const double* __restrict__ cand_hi,
                                                        const double* __restrict__ pelev_t, const int32_t* __restrict__ npath,
                                                        uint8_t* __restrict__ flags) {
  const int x = blockIdx.x;
  const int y = blockIdx.y * blockDim.x + threadIdx.x;
  if (y >= f.h) return;
  const size_t p = (size_t)y * f.wl + x;
  const int nc = ncand[x];
  if (nc < 0) {
    flags[p] = 1;
    return;
  }
  int n = npath[y];
  n = n < f.n_t ? n : f.n_t; // Iterator::zip, fast.rs:59-62
  const size_t hh = (size_t)f.h;
  const double step = f.p.simulation_step;
  bool flag = false;
  for (int q = 0; q < nc && !flag; q++) {
    const ObjectDev& o = f.objects[ccand[(size_t)x * COL_CAND + q]];
    const double lo = cand_lo[(size_t)x * COL_CAND + q], hi = cand_hi[(size_t)x * COL_CAND + q];
    // samples whose distance xs[i] (= i * step up to rounding) can lie in [lo, hi], one more on either side; the pairs (i - 1, i)
    // that contain one of them
    double a = dm_floor(lo / step) - 1.0, b = dm_floor(hi / step) + 2.0;
    if (!(a >= 1.0)) a = 1.0;           // also NaN / -inf
    if (!(b <= (double)(n - 1))) b = (double)(n - 1);
    const int i0 = (int)a, i1 = (int)b + 1 < n ? (int)b + 1 : n - 1;
    for (int i = i0; i <= i1; i++) {
      const double re0 = pelev_t[(size_t)(i - 1) * hh + y], re1 = pelev_t[(size_t)i * hh + y];
      if (!((re0 < o.vlo && re1 < o.vlo) || (re0 > o.vhi && re1 > o.vhi))) {
        flag = true;
        break;
      }
    }
  }
  flags[p] = flag ? 1 : 0;
}

// TerrainData::from_lat_lon's proximity filter (utils.rs:74-80) for every (sample, column): count, then fill ascending lists
template <bool FILL>
__global__ __launch_bounds__(256) void k_close_objects(Frame f, const double* __restrict__ plat,
                                                       const double* __restrict__ plon, const int32_t* __restrict__ ccand,
                                                       const int32_t* __restrict__ ncand, uint32_t* __restrict__ ccount,
                                                       const uint64_t* __restrict__ coffset, uint32_t* __restrict__ clist) {
  size_t n = (size_t)f.n_t * f.wl;
  size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= n) return;
  const int x = (int)(s % (size_t)f.wl);
  const int nc = ncand[x];
  const int32_t* cand = ccand + (size_t)x * COL_CAND;
  if (nc == 0) { // nothing can be close to this column
    if (!FILL) ccount[s] = 0;
    return;
  }
  const LatLonTrig t = latlon_trig(f.earth, plat[s], plon[s]);
  uint32_t c = 0;
  uint64_t k = FILL ? coffset[s] : 0;
  const int total = nc < 0 ? f.n_objects : nc;
  for (int q = 0; q < total; q++) {
    const int j = nc < 0 ? q : cand[q];
    if (object_is_close(f.earth, f.objects[j], t)) {
      if (FILL) clist[k + c] = (uint32_t)j;
      c++;
    }
  }
  if (!FILL) ccount[s] = c;
}

// Fast, phase C, general (scene objects: the full get_single_pixel, utils.rs:201-289).
// One image COLUMN per wavefront, lanes = 64 adjacent rows.  Everything that depends on the column only — the terrain
// profile value of a sample, its close-object list — is wave-uniform: scalar loads, and the "does this sample pair have close
// objects" test is a scalar branch, so the object logic runs for whole wavefronts at the 2 % of cells that need it and the
// other steps are a handful of vector instructions per row (one coalesced load of the ray elevations, which k_paths_transpose
// lays out sample-major for this kernel).  Inside an object cell a lane joins the geometry only if its segment enters the
// height band of an object (object_out_of_band).
#ifndef ATMRT_FAST_TRACE_WAVES
#define ATMRT_FAST_TRACE_WAVES 3
#endif
template <bool FILL>
__global__ __launch_bounds__(256, ATMRT_FAST_TRACE_WAVES) void k_fast_trace(Frame f, const double* __restrict__ prof,
                                                    const double* __restrict__ plat, const double* __restrict__ plon,
                                                    const uint32_t* __restrict__ ccount,
                                                    const uint64_t* __restrict__ coffset,
                                                    const uint32_t* __restrict__ clist, const double* __restrict__ pelev_t,
                                                    const double* __restrict__ plen_t, const int32_t* __restrict__ npath,
                                                    uint32_t* __restrict__ hit_count,
                                                    const uint64_t* __restrict__ hit_offset, PackedHits packed,
                                                    uint32_t* __restrict__ list_step, uint32_t* __restrict__ list_pixel,
                                                    uint32_t* __restrict__ px_steps,
                                                    unsigned long long* __restrict__ counters, double* __restrict__ step_prop,
                                                    const uint8_t* __restrict__ traced) {
  // traced (counting pass): only the rows marked by k_fast_flag_rows are traced here — the others cannot meet an object and keep
  // what the plain intersect scan found; a wavefront without a marked row leaves at once.
  // FILL = false: count the trace points of every pixel and keep those of pixels with <= RECT_SLOTS of them in the slot arena
  // (packed / list_step / list_pixel then are that arena, entry p * RECT_SLOTS + j).  FILL = true: write every point at its
  // place in the pixel-ordered list — for the pixels that did not fit their slots (hit_count > RECT_SLOTS); the others were
  // moved by k_fast_gather_trace_slots and their lanes are idle here (most wavefronts leave at once).
  // caches written by earlier kernels, read-only here: through the constant address space a wave-uniform index is a scalar load
  typedef const __attribute__((address_space(4))) double* ConstF64;
  typedef const __attribute__((address_space(4))) uint32_t* ConstU32;
  typedef const __attribute__((address_space(4))) uint64_t* ConstU64;
  typedef const __attribute__((address_space(4))) ObjectDev* ConstObj;
  const ConstF64 kprof = (ConstF64)(uintptr_t)prof, kplat = (ConstF64)(uintptr_t)plat, kplon = (ConstF64)(uintptr_t)plon;
  const ConstU32 kccount = (ConstU32)(uintptr_t)ccount, kclist = (ConstU32)(uintptr_t)clist;
  const ConstU64 kcoffset = (ConstU64)(uintptr_t)coffset;
  const ConstObj kobjects = (ConstObj)(uintptr_t)f.objects;
  const int lane = threadIdx.x & 63;
  const int x = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + (threadIdx.x >> 6)); // wave-uniform column
  const int y = blockIdx.y * 64 + lane;
  if (x >= f.wl) return; // whole wavefront
  const bool row_ok = y < f.h;
  const size_t hh = (size_t)f.h;
  const bool terrain_opaque = f.p.terrain_alpha == 1.0;
  int n = 0;
  if (row_ok) {
    n = npath[y];
    n = n < f.n_t ? n : f.n_t; // Iterator::zip, fast.rs:59-62
  }
  int nmax = n;
  for (int o = 32; o; o >>= 1) {
    const int v = __shfl_xor(nmax, o, 64);
    nmax = v > nmax ? v : nmax;
  }
  const size_t p = (size_t)(row_ok ? y : 0) * f.wl + x;
  uint64_t k = (FILL && row_ok) ? hit_offset[p] : 0;
  unsigned count = 0, stp = 0;
  const bool mine = FILL || !traced || (row_ok && traced[p] != 0);
  bool active = n > 1 && mine && (!FILL || hit_count[p] > (uint32_t)RECT_SLOTS);
  if (!__any(mine)) return;
  double te0 = kprof[x], re0 = n > 0 ? pelev_t[y] : 0.0;
  uint32_t c0 = kccount[x];
  constexpr int TCH = 8; // samples fetched ahead: every step's scalar loads would otherwise be a dependent round trip
  int i = 1;
  while (i < nmax) {
    if (!__any(active)) break;
    if (c0 == 0 && i + TCH <= nmax) {
      double te[TCH], re[TCH];
      uint32_t cc = 0;
#pragma unroll
      for (int q = 0; q < TCH; q++) {
        te[q] = kprof[(size_t)(i + q) * f.wl + x];
        cc |= kccount[(size_t)(i + q) * f.wl + x];
        re[q] = i + q < n ? pelev_t[(size_t)(i + q) * hh + y] : 0.0;
      }
      if (cc == 0) { // no close objects anywhere in the chunk: terrain only, utils.rs:222-240
#pragma unroll
        for (int q = 0; q < TCH; q++) {
          active = active && i + q < n;
          stp += active ? 1u : 0u;
          const double diff1 = re0 - te0, diff2 = re[q] - te[q];
          if (active && diff1 * diff2 < 0.0) {
            if (FILL || count < (unsigned)RECT_SLOTS) {
              const uint64_t kw = FILL ? k : (uint64_t)p * RECT_SLOTS + count;
              list_step[kw] = (uint32_t)(i + q - 1);
              list_pixel[kw] = (uint32_t)p;
              packed.color_tag[kw] = ATMRT_COLOR_TERRAIN;
              k++;
            }
            count++;
            if (terrain_opaque) active = false;
          }
          te0 = te[q];
          re0 = re[q];
        }
        i += TCH;
        continue;
      }
    }
    { // one sample pair, with or without close objects
    const size_t s1 = (size_t)i * f.wl + x, s0 = s1 - f.wl;
    const double te1 = kprof[s1];
    const uint32_t c1 = kccount[s1];
    const bool in_path = i < n;
    const double re1 = in_path ? pelev_t[(size_t)i * hh + y] : 0.0;
    active = active && in_path;
    stp += active ? 1u : 0u;
    const double diff1 = re0 - te0, diff2 = re1 - te1;
    const bool hit = active && diff1 * diff2 < 0.0; // utils.rs:222
    if ((c0 | c1) == 0) { // wave-uniform: no close objects at either sample — terrain only (utils.rs:222-240)
      if (hit) {
        if (FILL || count < (unsigned)RECT_SLOTS) {
          const uint64_t kw = FILL ? k : (uint64_t)p * RECT_SLOTS + count;
          list_step[kw] = (uint32_t)(i - 1);
          list_pixel[kw] = (uint32_t)p;
          packed.color_tag[kw] = ATMRT_COLOR_TERRAIN;
          k++;
        }
        count++;
        if (terrain_opaque) active = false;
      }
    } else { // the same for every lane: the union of the two ascending close lists (utils.rs:241-280)
      const ConstU32 la = kclist + kcoffset[s0];
      const ConstU32 lb = kclist + kcoffset[s1];
      if (active) {
        StepHits sh;
        sh.n = 0;
        sh.finish = false;
        if (hit) {
          step_push(sh, diff1 / (diff1 - diff2), -1, nullptr);
          if (terrain_opaque) sh.finish = true;
        }
        const double lat0 = kplat[s0], lon0 = kplon[s0], lat1 = kplat[s1], lon1 = kplon[s1];
        bool have_pos = false;
        Vec3 pos1 = v3(0.0, 0.0, 0.0), pos2 = pos1;
        // the union of the two ascending close lists, for the lanes whose segment enters the object's height band
        auto for_each_object = [&](auto&& visit) {
          uint32_t ia = 0, ib = 0;
          while (ia < c0 || ib < c1) {
            uint32_t idx;
            if (ib >= c1 || (ia < c0 && la[ia] <= lb[ib])) {
              idx = la[ia];
              if (ib < c1 && lb[ib] == idx) ib++;
              ia++;
            } else {
              idx = lb[ib++];
            }
            const double vlo = kobjects[idx].vlo, vhi = kobjects[idx].vhi;
            if ((re0 < vlo && re1 < vlo) || (re0 > vhi && re1 > vhi)) continue; // most rows pass above or below the object
            if (!have_pos) {
              pos1 = as_cartesian(f.earth, lat0, lon0, re0);
              pos2 = as_cartesian(f.earth, lat1, lon1, re1);
              have_pos = true;
            }
            visit((int)idx);
          }
        };
        for_each_object([&](int idx) { step_object(sh, f, idx, pos1, pos2); });
        const double d0 = i == 1 ? 0.0 : f.xs[i - 1], pl0 = i == 1 ? 0.0 : plen_t[(size_t)(i - 1) * hh + y];
        if (!FILL) {
          k = (uint64_t)p * RECT_SLOTS + count;
          if (sh.n > STEP_CANDIDATES) atomicAdd(&counters[6], 1ull); // the fill pass will need Workspace::step_prop
        }
        if (FILL && sh.n > STEP_CANDIDATES) { // big step: produce the points again, straight into the list, and sort them there
          const StepGeom g{lat0, lon0, re0, d0, pl0, lat1, lon1, re1, f.xs[i], plen_t[(size_t)i * hh + y]};
          const uint64_t k0 = k;
          if (hit) big_step_put(packed, step_prop, k++, diff1 / (diff1 - diff2), nullptr, g);
          for_each_object([&](int idx) { big_step_object(packed, step_prop, k, f, idx, pos1, pos2, g); });
          big_step_sort(packed, step_prop, k0, sh.n);
          for (uint64_t q = k0; q < k; q++) {
            list_step[q] = (uint32_t)(i - 1);
            list_pixel[q] = (uint32_t)p;
          }
        } else if (sh.n && (FILL || count + (unsigned)sh.n <= (unsigned)RECT_SLOTS)) {
          step_emit(sh, packed, list_step, list_pixel, k, (uint32_t)p, i - 1, lat0, lon0, re0, d0, pl0, lat1, lon1, re1, f.xs[i],
                    plen_t[(size_t)i * hh + y]);
        }
        count += (unsigned)sh.n;
        if (sh.finish) active = false;
      }
    }
    te0 = te1;
    re0 = re1;
    c0 = c1;

    }
    i++;
  }
  if (!FILL && row_ok && mine) {
    hit_count[p] = count;
    if (px_steps) px_steps[p] = stp;
  }
  if (!FILL) {
    unsigned long long steps = wave_sum((unsigned long long)(row_ok && mine ? stp : 0u));
    if (lane == 0 && steps) atomicAdd(&counters[0], steps);
  }
}

// Trace points kept in the slot arena by the counting pass of k_fast_trace, moved to their places in the pixel-ordered list
// (object points are complete; terrain points are completed by k_fast_finalize_list from list_step / list_pixel).
__global__ __launch_bounds__(256) void k_fast_gather_trace_slots(Frame f, const uint32_t* __restrict__ hit_count,
                                                                 const uint64_t* __restrict__ hit_offset,
                                                                 const uint32_t* __restrict__ slot_step, PackedHits sp,
                                                                 uint32_t* __restrict__ list_step, uint32_t* __restrict__ list_pixel,
                                                                 PackedHits packed) {
  const size_t plane = (size_t)f.wl * f.h;
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  const uint32_t n = hit_count[p];
  if (n > (uint32_t)RECT_SLOTS) return; // written by the fill pass
  const uint64_t k0 = hit_offset[p];
  for (uint32_t j = 0; j < n; j++) {
    const size_t q = p * RECT_SLOTS + j;
    const uint64_t k = k0 + j;
    list_step[k] = slot_step[q];
    list_pixel[k] = (uint32_t)p;
    const uint32_t tag = sp.color_tag[q];
    packed.color_tag[k] = tag;
    if (tag != ATMRT_COLOR_TERRAIN) {
      packed.lat[k] = sp.lat[q];
      packed.lon[k] = sp.lon[q];
      packed.distance[k] = sp.distance[q];
      packed.elevation[k] = sp.elevation[q];
      packed.path_length[k] = sp.path_length[q];
      for (int c = 0; c < 3; c++) packed.normal[3 * k + c] = sp.normal[3 * q + c];
      for (int c = 0; c < 4; c++) packed.rgba[4 * k + c] = sp.rgba[4 * q + c];
    }
  }
}

// pelev / plen [h][n_path_cap] -> [n_path_cap][h] for k_fast_trace (lanes = rows)
__global__ __launch_bounds__(256) void k_paths_transpose(Frame f, const double* __restrict__ pelev, const double* __restrict__ plen,
                                                         const int32_t* __restrict__ npath, double* __restrict__ pelev_t,
                                                         double* __restrict__ plen_t) {
  const int y = blockIdx.y * blockDim.x + threadIdx.x; // rows on grid.y (h < 32768), samples on grid.x (up to 4e6)
  const int i = blockIdx.x;
  if (y >= f.h) return;
  const bool have = i < npath[y]; // entries past the end of a row's path were never written
  const size_t src = (size_t)y * f.n_path_cap + i, dst = (size_t)i * f.h + y;
  pelev_t[dst] = have ? pelev[src] : 0.0;
  plen_t[dst] = have ? plen[src] : 0.0;
}

// ---------------------------------------------------------------------------------------------
// exclusive scan of hit_count (u32) into hit_offset (u64): block sums -> one-block scan -> apply
// ---------------------------------------------------------------------------------------------
constexpr int SCAN_ITEMS = 8; // per thread
__global__ __launch_bounds__(256) void k_scan_block_sums(const uint32_t* __restrict__ in, size_t n,
                                                         uint64_t* __restrict__ block_sums) {
  __shared__ unsigned long long sh[4];
  size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * SCAN_ITEMS;
  unsigned long long v = 0;
  for (int k = 0; k < SCAN_ITEMS; k++)
    if (base + k < n) v += in[base + k];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) block_sums[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ __launch_bounds__(256) void k_scan_sums(uint64_t* __restrict__ block_sums, size_t n_blocks,
                                                   unsigned long long* __restrict__ counters) {
  __shared__ unsigned long long sh[256];
  size_t per = (n_blocks + 255) / 256;
  size_t b0 = (size_t)threadIdx.x * per, b1 = b0 + per < n_blocks ? b0 + per : n_blocks;
  unsigned long long v = 0;
  for (size_t b = b0; b < b1; b++) v += block_sums[b];
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) { // Hillis–Steele inclusive scan
    unsigned long long t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned long long run = sh[threadIdx.x] - v;
  for (size_t b = b0; b < b1; b++) {
    unsigned long long t = block_sums[b];
    block_sums[b] = run;
    run += t;
  }
  if (threadIdx.x == 255) counters[1] = sh[255];
}
__global__ __launch_bounds__(256) void k_scan_apply(const uint32_t* __restrict__ in, size_t n,
                                                    const uint64_t* __restrict__ block_sums,
                                                    uint64_t* __restrict__ out) {
  __shared__ unsigned long long sh[256];
  size_t base = ((size_t)blockIdx.x * 256 + threadIdx.x) * SCAN_ITEMS;
  unsigned vals[SCAN_ITEMS];
  unsigned long long v = 0;
  for (int k = 0; k < SCAN_ITEMS; k++) {
    vals[k] = base + k < n ? in[base + k] : 0u;
    v += vals[k];
  }
  sh[threadIdx.x] = v;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    unsigned long long t = threadIdx.x >= (unsigned)off ? sh[threadIdx.x - off] : 0;
    __syncthreads();
    sh[threadIdx.x] += t;
    __syncthreads();
  }
  unsigned long long run = block_sums[blockIdx.x] + sh[threadIdx.x] - v;
  for (int k = 0; k < SCAN_ITEMS; k++) {
    if (base + k < n) out[base + k] = run;
    run += vals[k];
  }
}

// opaque mode: dense first hits -> packed list (pixel order)
__global__ __launch_bounds__(256) void k_pack_first_hits(Frame f, const uint64_t* __restrict__ hit_offset,
                                                         DensePlanes d, PackedHits packed) {
  size_t plane = (size_t)f.wl * f.h;
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane || d.hit_count[p] == 0) return;
  TracePointDev tp;
  tp.lat = d.lat[p];
  tp.lon = d.lon[p];
  tp.distance = d.distance[p];
  tp.elevation = d.elevation[p];
  tp.path_length = d.path_length[p];
  tp.normal = v3(d.normal[p], d.normal[plane + p], d.normal[2 * plane + p]);
  store_packed(packed, hit_offset[p], tp, f.p.terrain_alpha);
}


// ---------------------------------------------------------------------------------------------
// InterpolatingRectilinear (interpolating_rectilinear.rs): pinhole ray table -> angular lattice ->
// the lattice frame goes through the Fast pipeline above -> 4-corner blend per pixel.
// ---------------------------------------------------------------------------------------------

// gen_fov_data :456-468: ray_params_table for the FULL image (the lattice steps are global minima)
__global__ __launch_bounds__(256) void k_fov_table(Frame f, double* __restrict__ dir, double* __restrict__ elev) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y;
  if (x >= f.p.width) return;
  double d, e;
  rect_ray_params(f.p, f.ph, x, y, d, e);
  dir[(size_t)y * f.p.width + x] = d;
  elev[(size_t)y * f.p.width + x] = e;
}
// min_elev_step per column :470-491 and min_dir_step per row :493-517 (before the global min and * SCALE)
// Both are minima over a whole column / row; they are folded in parallel (the values are positive, so their bit patterns order
// like the numbers and an integer atomicMin merges the partial results exactly).
constexpr int FOV_ROW_CHUNKS = 32;
__global__ __launch_bounds__(256) void k_fov_colmin_init(Frame f, double* __restrict__ colmin) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x < f.p.width) colmin[x] = dm_to_radians(360.0);
}
__global__ __launch_bounds__(256) void k_fov_colmin(Frame f, const double* __restrict__ elev, double* __restrict__ colmin) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  if (x >= f.p.width) return;
  const int W = f.p.width, H = f.p.height;
  const int per = (H - 1 + FOV_ROW_CHUNKS - 1) / FOV_ROW_CHUNKS;
  const int y0 = 1 + blockIdx.y * per, y1 = y0 + per < H ? y0 + per : H; // this block's diffs: rows y0 .. y1 - 1 against y - 1
  if (y0 >= y1) return;
  const double min_diff = dm_to_radians(f.p.frame.fov) / (double)f.p.width / 3.0;
  double mn = dm_to_radians(360.0), last = elev[(size_t)(y0 - 1) * W + x];
  for (int y = y0; y < y1; y++) {
    double next = elev[(size_t)y * W + x];
    double diff = dm_fabs(next - last);
    if (diff < min_diff) diff = min_diff;
    if (diff < mn) mn = diff;
    last = next;
  }
  atomicMin((unsigned long long*)&colmin[x], (unsigned long long)dm_bits(mn));
}
__global__ __launch_bounds__(256) void k_fov_rowmin(Frame f, const double* __restrict__ dir, double* __restrict__ rowmin) {
  const int y = blockIdx.x * 4 + (threadIdx.x >> 6); // one row per wavefront
  const int lane = threadIdx.x & 63;
  if (y >= f.p.height) return;
  const int W = f.p.width;
  const double full = dm_to_radians(360.0);
  const double min_diff = dm_to_radians(f.p.frame.fov) / (double)f.p.width / 3.0;
  const double* row = dir + (size_t)y * W;
  double mn = full;
  for (int x = 1 + lane; x < W; x += 64) {
    double diff = dm_fabs(row[x] - row[x - 1]);
    if (diff > full) diff -= full;
    if (diff < min_diff) diff = min_diff;
    if (diff < mn) mn = diff;
  }
  for (int o = 32; o; o >>= 1) {
    const double v = __shfl_xor(mn, o, 64);
    mn = v < mn ? v : mn;
  }
  if (lane == 0) rowmin[y] = mn;
}

static __device__ __forceinline__ int sat_i32(double v) { // Rust `f as i32`
  if (v != v) return 0;
  if (v <= -2147483648.0) return (int)0x80000000;
  if (v >= 2147483647.0) return 2147483647;
  return (int)v;
}

// FovData::cache_coords :186-204 for every pixel of the shard + bounds of the referenced lattice.  Grid-stride over
// pixels; each block folds its bounds through LDS and issues four atomics (one word per bound would otherwise see one
// atomic per wavefront: 5.9 ms at the headline size, now 0.1 ms).
__global__ __launch_bounds__(256) void k_lattice_keys(Frame f, InterpBuffers ib, double min_elev_step, double min_dir_step) {
  __shared__ int sh[4][4];
  const size_t npx = (size_t)f.wl * f.h;
  int ei = 0x7fffffff, di = 0x7fffffff, ej = (int)0x80000000, dj = (int)0x80000000;
  for (size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x; p < npx; p += (size_t)gridDim.x * blockDim.x) {
    const int y = (int)(p / (size_t)f.wl), x = (int)(p % (size_t)f.wl);
    const size_t src = (size_t)y * f.p.width + f.c0 + x;
    const double ef = ib.elev[src] / min_elev_step, df = ib.dir[src] / min_dir_step;
    const int e = sat_i32(dm_floor(ef)), d = sat_i32(dm_floor(df));
    ib.key_e[p] = e;
    ib.key_d[p] = d;
    ib.rem_e[p] = ef - (double)e;
    ib.rem_d[p] = df - (double)d;
    ei = e < ei ? e : ei;
    ej = e > ej ? e : ej;
    di = d < di ? d : di;
    dj = d > dj ? d : dj;
  }
  for (int off = 32; off > 0; off >>= 1) {
    int a = __shfl_down(ei, off, 64), b = __shfl_down(di, off, 64), c = __shfl_down(ej, off, 64), d = __shfl_down(dj, off, 64);
    ei = a < ei ? a : ei;
    di = b < di ? b : di;
    ej = c > ej ? c : ej;
    dj = d > dj ? d : dj;
  }
  const int wave = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    sh[wave][0] = ei; sh[wave][1] = ej; sh[wave][2] = di; sh[wave][3] = dj;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < 4; w++) {
      ei = sh[w][0] < ei ? sh[w][0] : ei;
      ej = sh[w][1] > ej ? sh[w][1] : ej;
      di = sh[w][2] < di ? sh[w][2] : di;
      dj = sh[w][3] > dj ? sh[w][3] : dj;
    }
    if (ei != 0x7fffffff) {
      atomicMin(&ib.bounds[0], ei);
      atomicMax(&ib.bounds[1], ej);
      atomicMin(&ib.bounds[2], di);
      atomicMax(&ib.bounds[3], dj);
    }
  }
}

// TracePoint incl. colour, generators/mod.rs:21-30
struct FullTP {
  double lat, lon, distance, elevation, path_length;
  Vec3 normal;
  uint32_t tag;
  double rgba[4];
};
static __device__ __forceinline__ FullTP load_tp(const PackedHits& h, uint64_t k) {
  FullTP t;
  t.lat = h.lat[k];
  t.lon = h.lon[k];
  t.distance = h.distance[k];
  t.elevation = h.elevation[k];
  t.path_length = h.path_length[k];
  t.normal = v3(h.normal[3 * k], h.normal[3 * k + 1], h.normal[3 * k + 2]);
  t.tag = h.color_tag[k];
  for (int q = 0; q < 4; q++) t.rgba[q] = h.rgba[4 * k + q];
  return t;
}
static __device__ __forceinline__ void store_tp(const PackedHits& h, uint64_t k, const FullTP& t) {
  h.lat[k] = t.lat;
  h.lon[k] = t.lon;
  h.distance[k] = t.distance;
  h.elevation[k] = t.elevation;
  h.path_length[k] = t.path_length;
  h.normal[3 * k] = t.normal.x;
  h.normal[3 * k + 1] = t.normal.y;
  h.normal[3 * k + 2] = t.normal.z;
  h.color_tag[k] = t.tag;
  for (int q = 0; q < 4; q++) h.rgba[4 * k + q] = t.rgba[q];
}
// TracePoint::interpolate (generators/mod.rs:33-43) + PixelColor::interpolate (:67-79)
static __device__ FullTP tp_interpolate(const FullTP& a, const FullTP& b, double c) {
  FullTP r;
  r.lat = a.lat * (1.0 - c) + b.lat * c;
  r.lon = a.lon * (1.0 - c) + b.lon * c;
  r.distance = a.distance * (1.0 - c) + b.distance * c;
  r.elevation = a.elevation * (1.0 - c) + b.elevation * c;
  r.path_length = a.path_length * (1.0 - c) + b.path_length * c;
  r.normal = v3(a.normal.x * (1.0 - c) + b.normal.x * c, a.normal.y * (1.0 - c) + b.normal.y * c,
                a.normal.z * (1.0 - c) + b.normal.z * c);
  if (a.tag == ATMRT_COLOR_TERRAIN && b.tag == ATMRT_COLOR_TERRAIN) {
    r.tag = ATMRT_COLOR_TERRAIN;
    r.rgba[0] = r.rgba[1] = r.rgba[2] = 0.0;
    r.rgba[3] = a.rgba[3] * (1.0 - c) + b.rgba[3] * c;
  } else if (a.tag == ATMRT_COLOR_RGBA && b.tag == ATMRT_COLOR_RGBA) {
    r.tag = ATMRT_COLOR_RGBA;
    for (int q = 0; q < 4; q++) r.rgba[q] = a.rgba[q] * (1.0 - c) + b.rgba[q] * c;
  } else {
    r.tag = ATMRT_COLOR_TERRAIN;
    r.rgba[0] = r.rgba[1] = r.rgba[2] = 0.0;
    r.rgba[3] = a.tag == ATMRT_COLOR_TERRAIN ? a.rgba[3] : b.rgba[3];
  }
  return r;
}

// Trace points of the four corner pixels together that the in-register member list holds; pixels with more are blended by
// k_interp_blend_big over a member arena in HBM.  4 = no middle instance: a 64-member in-register instance (round 1) cost 230-274
// VGPRs and 0.5-1 KB of scratch for its arrays, the arena instance needs 48-148 VGPRs and none — interpolating frame at
// terrain_alpha 0.5: 33.3 ms with 64, 26.5 (32), 20.7 (16), 20.1 ms with 4; config 5: 67.6 -> 53.5 ms.
#ifndef ATMRT_INTERP_MEMBERS
#define ATMRT_INTERP_MEMBERS 4
#endif
constexpr int INTERP_MEMBERS = ATMRT_INTERP_MEMBERS;

// interpolate_trace_points :267-337 on the group's corner members (index -1 = None)
static __device__ bool interp_group(const LatticeResult& lr, const uint64_t* member_k, const int e[4], double re, double rd,
                                    FullTP& out) {
  int mask = (e[0] >= 0 ? 1 : 0) | (e[1] >= 0 ? 2 : 0) | (e[2] >= 0 ? 4 : 0) | (e[3] >= 0 ? 8 : 0);
  auto tp = [&](int c) { return load_tp(lr.hits, member_k[e[c]]); };
  auto two_adjacent = [&](int c0, int c1, double rem_elev, double rem_dir) { // :339-350
    if (rem_elev >= 0.5) return false;
    out = tp_interpolate(tp(c0), tp(c1), rem_dir);
    return true;
  };
  auto two_diagonal = [&](int c0, int c1, double rem_elev, double rem_dir) { // :352-364
    if ((rem_elev >= 0.5 && rem_dir < 0.5) || (rem_elev < 0.5 && rem_dir >= 0.5)) return false;
    double coeff = rem_elev * rem_dir / (rem_elev * rem_dir + (1.0 - rem_elev) * (1.0 - rem_dir));
    out = tp_interpolate(tp(c0), tp(c1), coeff);
    return true;
  };
  auto three = [&](int c0, int c1, int c2, double rem_elev, double rem_dir) { // :366-380
    if (rem_elev >= 0.5 && rem_dir >= 0.5) return false;
    double sum = 1.0 - rem_elev + rem_elev * (1.0 - rem_dir);
    FullTP in = tp_interpolate(tp(c0), tp(c1), rem_dir);
    out = tp_interpolate(in, tp(c2), rem_elev * (1.0 - rem_dir) / sum);
    return true;
  };
  switch (mask) {
    case 0: return false;
    case 1: if (re < 0.5 && rd < 0.5) { out = tp(0); return true; } return false;
    case 2: if (re < 0.5 && rd >= 0.5) { out = tp(1); return true; } return false;
    case 4: if (re >= 0.5 && rd < 0.5) { out = tp(2); return true; } return false;
    case 8: if (re >= 0.5 && rd >= 0.5) { out = tp(3); return true; } return false;
    case 1 | 2: return two_adjacent(0, 1, re, rd);
    case 1 | 4: return two_adjacent(0, 2, rd, re);
    case 1 | 8: return two_diagonal(0, 3, re, rd);
    case 2 | 4: return two_diagonal(1, 2, re, 1.0 - rd);
    case 2 | 8: return two_adjacent(1, 3, 1.0 - rd, re);
    case 4 | 8: return two_adjacent(2, 3, 1.0 - re, rd);
    case 1 | 2 | 4: return three(0, 1, 2, re, rd);
    case 1 | 2 | 8: return three(1, 0, 3, re, 1.0 - rd);
    case 1 | 4 | 8: return three(0, 3, 2, 1.0 - re, rd);
    case 2 | 4 | 8: return three(3, 2, 1, 1.0 - re, 1.0 - rd);
    default: {
      FullTP i1 = tp_interpolate(tp(0), tp(1), rd);
      FullTP i2 = tp_interpolate(tp(2), tp(3), rd);
      out = tp_interpolate(i1, i2, re);
      return true;
    }
  }
}

// collect_trace_points :213-243 + match_sequence :245-265 + interpolate_trace_points per group for one pixel: the members
// (trace points of the four corners, corner by corner) are grouped — a point joins the first group, in creation order, that
// holds a point of its class within step_size of its distance — and every group is blended.  Returns the number of trace
// points of the pixel; FILL writes them at k_out.  The member arrays hold at least the corners' total number of points.
template <bool FILL, class GroupT>
static __device__ __forceinline__ unsigned blend_members(const LatticeResult& lr, const size_t corner[4], double rem_elev,
                                                         double rem_dir, double step_size, uint64_t* member_k, double* member_dist,
                                                         uint8_t* member_corner, uint8_t* member_tag, GroupT* member_group,
                                                         uint64_t k_out, const PackedHits& packed) {
  int n_members = 0, n_groups = 0;
  for (int c = 0; c < 4; c++) {
    uint64_t k0 = lr.hit_offset[corner[c]];
    uint32_t cnt = lr.hit_count[corner[c]];
    for (uint32_t q = 0; q < cnt; q++) {
      double dist = lr.hits.distance[k0 + q];
      uint8_t tag = (uint8_t)lr.hits.color_tag[k0 + q];
      int found = -1;
      for (int g = 0; g < n_groups && found < 0; g++) // first group (creation order) with any close point of the same class
        for (int m = 0; m < n_members; m++)
          if ((int)member_group[m] == g && dm_fabs(dist - member_dist[m]) < step_size && tag == member_tag[m]) {
            found = g;
            break;
          }
      if (found < 0) found = n_groups++;
      member_k[n_members] = k0 + q;
      member_dist[n_members] = dist;
      member_corner[n_members] = (uint8_t)c;
      member_tag[n_members] = tag;
      member_group[n_members] = (GroupT)found;
      n_members++;
    }
  }
  unsigned count = 0;
  for (int g = 0; g < n_groups; g++) {
    int e[4] = {-1, -1, -1, -1};
    for (int m = 0; m < n_members; m++)
      if ((int)member_group[m] == g) e[member_corner[m]] = m; // later entries overwrite, :247-263
    FullTP tp;
    if (interp_group(lr, member_k, e, rem_elev, rem_dir, tp)) {
      if (FILL) store_tp(packed, k_out + count, tp);
      count++;
    }
  }
  return count;
}

// interpolate :395-418 with collect_trace_points :213-243 and match_sequence :245-265
// CAP = 4: the pixels whose four lattice corners hold at most four trace points together (nearly all of them) — the member
// arrays are four registers wide; it also writes what does not depend on the members (referenced marks, blended angles) and sizes
// the arena of k_interp_blend_big, which blends the others.  (A middle instance, CAP = INTERP_MEMBERS > 4, exists in the template
// but is not launched: see INTERP_MEMBERS.)
template <bool FILL, int CAP>
__global__ __launch_bounds__(256) void k_interp_blend(Frame f, InterpBuffers ib, LatticeResult lr, DensePlanes out,
                                                      const uint64_t* __restrict__ hit_offset, PackedHits packed,
                                                      unsigned long long* __restrict__ counters) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y;
  if (x >= f.wl) return;
  const size_t p = (size_t)y * f.wl + x;
  const double rem_elev = ib.rem_e[p], rem_dir = ib.rem_d[p];
  const double step_size = f.p.simulation_step;
  size_t corner[4]; // SEQUENCE = (0,0) (0,1) (1,0) (1,1): (elev_index + i, dir_index + j)
  for (int s = 0; s < 4; s++)
    corner[s] = (size_t)(ib.key_e[p] + (s >> 1) - f.ei0) * lr.nd + (size_t)(ib.key_d[p] + (s & 1) - f.di0);
  uint32_t total = 0;
  for (int c = 0; c < 4; c++) {
    if (!FILL && CAP == 4) ib.referenced[corner[c]] = 1;
    total += lr.hit_count[corner[c]];
  }
  if (!FILL && CAP == 4) {
    double e0 = lr.elevation_angle[corner[0]], e1 = lr.elevation_angle[corner[1]], e2 = lr.elevation_angle[corner[2]],
           e3 = lr.elevation_angle[corner[3]];
    double a0 = lr.azimuth[corner[0]], a1 = lr.azimuth[corner[1]], a2 = lr.azimuth[corner[2]], a3 = lr.azimuth[corner[3]];
    out.elevation_angle[p] = e0 * (1.0 - rem_elev) * (1.0 - rem_dir) + e1 * (1.0 - rem_elev) * rem_dir +
                             e2 * rem_elev * (1.0 - rem_dir) + e3 * rem_elev * rem_dir;
    out.azimuth[p] = a0 * (1.0 - rem_elev) * (1.0 - rem_dir) + a1 * (1.0 - rem_elev) * rem_dir + a2 * rem_elev * (1.0 - rem_dir) +
                     a3 * rem_elev * rem_dir;
  }
  if (!FILL && CAP == 4 && total > (uint32_t)INTERP_MEMBERS) { // k_interp_blend_big's pixel: size its member arena
    atomicAdd(&counters[7], 1ull);
    atomicAdd(&counters[8], (unsigned long long)total);
  }
  if (CAP == 4 ? total > 4u : (total <= 4u || total > (uint32_t)INTERP_MEMBERS)) return; // another instance's pixel
  uint64_t member_k[CAP];
  double member_dist[CAP];
  uint8_t member_corner[CAP], member_tag[CAP], member_group[CAP];
  const unsigned count = blend_members<FILL>(lr, corner, rem_elev, rem_dir, step_size, member_k, member_dist, member_corner, member_tag,
                                             member_group, FILL ? hit_offset[p] : 0, packed);
  if (!FILL) out.hit_count[p] = count;
}

// The pixels whose four lattice corners hold more than INTERP_MEMBERS trace points together (the reference's Vec has no bound,
// interpolating_rectilinear.rs:213-243): same grouping and blend, member list in an HBM arena handed out by an atomic cursor
// (each pixel's range is private, so the order in which pixels take their ranges does not matter).
template <bool FILL>
__global__ __launch_bounds__(256) void k_interp_blend_big(Frame f, InterpBuffers ib, LatticeResult lr, DensePlanes out,
                                                          const uint64_t* __restrict__ hit_offset, PackedHits packed,
                                                          BlendArena arena, unsigned long long* __restrict__ counters) {
  int x = blockIdx.x * blockDim.x + threadIdx.x;
  int y = blockIdx.y;
  if (x >= f.wl) return;
  const size_t p = (size_t)y * f.wl + x;
  size_t corner[4];
  uint32_t total = 0;
  for (int s = 0; s < 4; s++) {
    corner[s] = (size_t)(ib.key_e[p] + (s >> 1) - f.ei0) * lr.nd + (size_t)(ib.key_d[p] + (s & 1) - f.di0);
    total += lr.hit_count[corner[s]];
  }
  if (total <= (uint32_t)INTERP_MEMBERS) return;
  const unsigned long long base = atomicAdd(&counters[9], (unsigned long long)total);
  const unsigned count = blend_members<FILL>(lr, corner, ib.rem_e[p], ib.rem_d[p], f.p.simulation_step, arena.k + base, arena.dist + base,
                                             arena.corner + base, arena.tag + base, arena.group + base, FILL ? hit_offset[p] : 0, packed);
  if (!FILL) out.hit_count[p] = count;
}


// ray-steps of the lattice pixels the image actually references (the reference memoises exactly those); grid-stride,
// one atomic per block
__global__ __launch_bounds__(256) void k_lattice_steps(size_t n, const uint8_t* __restrict__ referenced,
                                                       const uint32_t* __restrict__ px_steps,
                                                       unsigned long long* __restrict__ counters) {
  __shared__ unsigned long long sh[4];
  unsigned long long v = 0;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    if (referenced[i]) v += px_steps[i];
  v = wave_sum(v);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = v;
  __syncthreads();
  if (threadIdx.x == 0) {
    v = sh[0] + sh[1] + sh[2] + sh[3];
    if (v) atomicAdd(&counters[0], v);
  }
}

void launch_fov_table(const Frame& f, const InterpBuffers& ib, hipStream_t stream) {
  hipLaunchKernelGGL(k_fov_table, dim3(cdiv(f.p.width, 256), f.p.height), dim3(256), 0, stream, f, ib.dir, ib.elev);
  hipLaunchKernelGGL(k_fov_colmin_init, dim3(cdiv(f.p.width, 256)), dim3(256), 0, stream, f, ib.colmin);
  hipLaunchKernelGGL(k_fov_colmin, dim3(cdiv(f.p.width, 256), FOV_ROW_CHUNKS), dim3(256), 0, stream, f, ib.elev, ib.colmin);
  hipLaunchKernelGGL(k_fov_rowmin, dim3(cdiv(f.p.height, 4)), dim3(256), 0, stream, f, ib.dir, ib.rowmin);
}
void launch_lattice_keys(const Frame& f, const InterpBuffers& ib, double min_elev_step, double min_dir_step, hipStream_t stream) {
  const size_t npx = (size_t)f.wl * f.h;
  hipLaunchKernelGGL(k_lattice_keys, dim3(npx < 1024 * 256 ? cdiv(npx, 256) : 1024), dim3(256), 0, stream, f, ib, min_elev_step,
                     min_dir_step);
}
void launch_interp_blend(const Frame& f, Workspace& ws, const InterpBuffers& ib, const LatticeResult& lr, bool fill,
                         const DensePlanes& dense, const PackedHits& packed, hipStream_t stream) {
  dim3 grid(cdiv(f.wl, 256), f.h);
  if (fill) {
    hipLaunchKernelGGL((k_interp_blend<true, 4>), grid, dim3(256), 0, stream, f, ib, lr, dense, ws.hit_offset, packed,
                       (unsigned long long*)ws.counters);
    if (INTERP_MEMBERS > 4)
      hipLaunchKernelGGL((k_interp_blend<true, INTERP_MEMBERS>), grid, dim3(256), 0, stream, f, ib, lr, dense, ws.hit_offset, packed,
                         (unsigned long long*)ws.counters);
  } else {
    hipLaunchKernelGGL((k_interp_blend<false, 4>), grid, dim3(256), 0, stream, f, ib, lr, dense, (const uint64_t*)nullptr, packed,
                       (unsigned long long*)ws.counters);
    if (INTERP_MEMBERS > 4)
      hipLaunchKernelGGL((k_interp_blend<false, INTERP_MEMBERS>), grid, dim3(256), 0, stream, f, ib, lr, dense, (const uint64_t*)nullptr,
                         packed, (unsigned long long*)ws.counters);
  }
}
void launch_interp_blend_big(const Frame& f, Workspace& ws, const InterpBuffers& ib, const LatticeResult& lr, bool fill,
                             const DensePlanes& dense, const PackedHits& packed, const BlendArena& arena, hipStream_t stream) {
  dim3 grid(cdiv(f.wl, 256), f.h);
  (void)hipMemsetAsync(ws.counters + 9, 0, sizeof(uint64_t), stream); // the arena cursor
  if (fill)
    hipLaunchKernelGGL((k_interp_blend_big<true>), grid, dim3(256), 0, stream, f, ib, lr, dense, ws.hit_offset, packed, arena,
                       (unsigned long long*)ws.counters);
  else
    hipLaunchKernelGGL((k_interp_blend_big<false>), grid, dim3(256), 0, stream, f, ib, lr, dense, (const uint64_t*)nullptr, packed,
                       arena, (unsigned long long*)ws.counters);
}
void launch_interp_finish(const Frame& f, Workspace& ws, const InterpBuffers& ib, const LatticeResult& lr,
                          const DensePlanes& dense, const PackedHits& packed, hipStream_t stream) {
  size_t n = (size_t)lr.nd * lr.ne;
  hipLaunchKernelGGL(k_lattice_steps, dim3(n < 1024 * 256 ? cdiv(n, 256) : 1024), dim3(256), 0, stream, n, ib.referenced, lr.px_steps,
                     (unsigned long long*)ws.counters);
  hipLaunchKernelGGL(k_dense_from_packed, dim3(cdiv(f.wl, 256), f.h), dim3(256), 0, stream, f, ws.hit_offset, packed, dense, 0);
}


// ---------------------------------------------------------------------------------------------
// renderer::draw_image (renderer/mod.rs:385-414) — front-to-back alpha compositing of each
// pixel's trace points with the colouring method and optional fog; 3 B per pixel leave the kernel instead of 88.
// PACKED: trace points from the packed lists; otherwise from the dense first-hit planes (opaque frames).
// ---------------------------------------------------------------------------------------------
template <bool PACKED>
__global__ __launch_bounds__(256) void k_draw_image(size_t n_pixels, atmrt_coloring_t col, double terrain_alpha,
                                                    const uint32_t* __restrict__ hit_count,
                                                    const uint64_t* __restrict__ hit_offset, PackedHits hits, DensePlanes dense,
                                                    uint8_t* __restrict__ rgb) {
  size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n_pixels) return;
  Rgb8 result{{0, 0, 0}};
  double accum_neg_alpha = 1.0;
  const uint32_t cnt = hit_count[p];
  const uint64_t k0 = PACKED ? hit_offset[p] : 0;
  for (uint32_t q = 0; q < cnt; q++) {
    double distance, elevation, path_length, nx, ny, nz, r = 0.0, g = 0.0, b = 0.0, alpha;
    uint32_t tag;
    if (PACKED) {
      uint64_t k = k0 + q;
      distance = hits.distance[k]; elevation = hits.elevation[k]; path_length = hits.path_length[k];
      nx = hits.normal[3 * k]; ny = hits.normal[3 * k + 1]; nz = hits.normal[3 * k + 2];
      tag = hits.color_tag[k];
      r = hits.rgba[4 * k]; g = hits.rgba[4 * k + 1]; b = hits.rgba[4 * k + 2]; alpha = hits.rgba[4 * k + 3];
    } else {
      distance = dense.distance[p]; elevation = dense.elevation[p]; path_length = dense.path_length[p];
      nx = dense.normal[p]; ny = dense.normal[n_pixels + p]; nz = dense.normal[2 * n_pixels + p];
      tag = ATMRT_COLOR_TERRAIN;
      alpha = terrain_alpha;
    }
    Rgb8 c1 = col.kind == ATMRT_COLORING_SIMPLE ? simple_color(col, distance, elevation)
                                                 : shading_color(col, nx, ny, nz, elevation, tag, r, g, b);
    Rgb8 c2 = col.has_fog ? apply_fog(col.fog_distance, path_length, c1) : c1;
    result = add_rgb(result, c2, accum_neg_alpha * alpha);
    accum_neg_alpha *= 1.0 - alpha;
  }
  result = add_rgb(result, default_color(col), accum_neg_alpha);
  rgb[3 * p] = result.c[0];
  rgb[3 * p + 1] = result.c[1];
  rgb[3 * p + 2] = result.c[2];
}

void launch_draw_image(size_t n_pixels, const atmrt_coloring_t& col, double terrain_alpha, bool packed_valid,
                       const uint32_t* hit_count, const uint64_t* hit_offset, const PackedHits& hits, const DensePlanes& dense,
                       uint8_t* rgb, hipStream_t stream) {
  if (packed_valid)
    hipLaunchKernelGGL((k_draw_image<true>), dim3(cdiv(n_pixels, 256)), dim3(256), 0, stream, n_pixels, col, terrain_alpha,
                       hit_count, hit_offset, hits, dense, rgb);
  else
    hipLaunchKernelGGL((k_draw_image<false>), dim3(cdiv(n_pixels, 256)), dim3(256), 0, stream, n_pixels, col, terrain_alpha,
                       hit_count, hit_offset, hits, dense, rgb);
}

// ---------------------------------------------------------------------------------------------
// harness kernels
// ---------------------------------------------------------------------------------------------
__global__ void k_get_elev(Frame f, size_t n, const double* lat, const double* lon, double* elev, uint8_t* valid) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double e = 0.0;
  bool ok = terrain_get_elev(f.tv, lat[i], lon[i], e);
  elev[i] = ok ? e : 0.0;
  valid[i] = ok ? 1 : 0;
}
__global__ void k_ray_paths(Frame f, double h0, size_t n_angles, const double* angles_deg, int straight, double step,
                            size_t n_steps, double* x, double* h) {
  size_t a = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (a >= n_angles) return;
  const bool sph = f.earth.spherical != 0;
  Stepper s;
  stepper_init(s, sph, f.earth.shape_radius, h0, dm_to_radians(angles_deg[a]));
  size_t base = a * (n_steps + 1);
  x[base] = 0.0;
  h[base] = h0;
  for (size_t k = 1; k <= n_steps; k++) {
    RayState st = stepper_next(s, *f.atm, sph, f.earth.shape_radius, straight != 0, step);
    x[base + k] = st.x;
    h[base + k] = st.h;
  }
}
__global__ void k_atm_sample(Frame f, size_t n, const double* alt, double* t, double* p, double* nidx, double* dn) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  t[i] = atm_temperature(*f.atm, alt[i]);
  p[i] = atm_pressure(*f.atm, alt[i]);
  // n and dn/dh through the steppers' own evaluation (the first call finds the lane's layer, the second then takes the certified
  // shortcut path when the whole wavefront sits in one certified interval, the IEEE path otherwise): same values as refr_n / refr_dn
  int hint = 0;
  double nv, dv;
  refr_n_dn_hint<true>(*f.atm, alt[i], hint, nv, dv);
  refr_n_dn_hint<true>(*f.atm, alt[i], hint, nv, dv);
  nidx[i] = nv;
  dn[i] = dv;
}
__global__ void k_coords_at_dist(Frame f, double lat0, double lon0, double dir, size_t n, const double* dist, double* lat,
                                 double* lon) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Earth e = f.earth;
  e.flat_dirs &= ~EARTH_FAST_DIV; // arbitrary distances here, not a stepper's
  DirCalc c;
  dircalc_new(e, lat0, lon0, dir, c);
  coords_at_dist(e, c, dist[i], lat[i], lon[i]);
}

// detmath.h element-wise (atmrt_math_probe): the GPU's instruction sequences against the host's on arbitrary operands
__global__ void k_math_probe(int op, size_t n, const double* __restrict__ a, const double* __restrict__ b,
                             double* __restrict__ out0, double* __restrict__ out1) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = a[i], y = b ? b[i] : 0.0;
  double r0 = 0.0, r1 = 0.0;
  switch (op) {
    case ATMRT_PROBE_DIV: r0 = dm_div(x, y); break;
    case ATMRT_PROBE_DIV_R: r0 = dm_div_r(x, y, 1.0 / y); break;
    case ATMRT_PROBE_SQRT_INRANGE: r0 = dm_sqrt_inrange(x); break;
    case ATMRT_PROBE_EXP: r0 = dm_exp(x); break;
    case ATMRT_PROBE_LOG: r0 = dm_log(x); break;
    case ATMRT_PROBE_POW: r0 = dm_pow(x, y); break;
    case ATMRT_PROBE_SINCOS: dm_sincos(x, &r0, &r1); break;
    case ATMRT_PROBE_ASIN: r0 = dm_asin(x); break;
    case ATMRT_PROBE_ATAN2: r0 = dm_atan2(x, y); break;
    case ATMRT_PROBE_IEEE_DIV: r0 = x / y; break;
    case ATMRT_PROBE_IEEE_SQRT: r0 = dm_sqrt(x); break;
    case ATMRT_PROBE_ATAN: r0 = dm_atan(x); break;
    case ATMRT_PROBE_TAN: r0 = dm_tan(x); break;
    case ATMRT_PROBE_POW3: { // the three-point form the stepping kernels use (wave votes on the range guards)
      double p1, p2;
      pow3(x, x * 0.99999981, x * 1.00000019, y, r0, p1, p2);
      r1 = p1 + p2;
      break;
    }
    case ATMRT_PROBE_DIV3: {
      double q0;
      dm_div3(x, y, x, y * 0.99999976158142090, x, y * 1.00000047683715820, &q0, &r0, &r1);
      break;
    }
    case ATMRT_PROBE_DIV3_SEEDED: {
      double q0;
      dm_div3_seeded(x, y, x, y * 0.99999976158142090, x, y * 1.00000023841857910, 0, 0.0, &q0, &r0, &r1);
      break;
    }
    case ATMRT_PROBE_DIV3_SEED_Z: {
      double q1;
      dm_div3_seeded(x, y, x, y * 0.99999976158142090, x, y * 1.00000023841857910, 1, 2.0 - y, &r0, &q1, &r1);
      break;
    }
    case ATMRT_PROBE_DIV_SEED_N: r0 = dm_div_seeded(x, 1.0 + y, 1.0 - y); break;
    case ATMRT_PROBE_POW3_SHARED: { // as refr_n_layer3 calls it on a tight segment (the threshold as atm_certify computes it)
      const double margin = 1.01 * (dm_fabs(y) * (4.76837158203125e-07 + 1.0e-14)) * DM_INVLN2N + 1.0e-6;
      double p1, p2;
      pow3_tight(x, x * 0.99999976158142090, x * 1.00000023841857910, y, margin < 0.5 ? 0.5 - margin : -1.0, r0, p1, p2);
      r1 = p1 + p2;
      break;
    }
    default: break;
  }
  out0[i] = r0;
  if (out1) out1[i] = r1;
}
void launch_math_probe(int op, size_t n, const double* a, const double* b, double* out0, double* out1, hipStream_t stream) {
  if (n) hipLaunchKernelGGL(k_math_probe, dim3(cdiv(n, 256)), dim3(256), 0, stream, op, n, a, b, out0, out1);
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------

void launch_resolve(const Frame& f, Workspace& ws, ObjectDev* objects_mut, hipStream_t stream) {
  hipLaunchKernelGGL(k_resolve, dim3(cdiv((size_t)f.n_objects + 1, 64)), dim3(64), 0, stream, f, ws.alt, objects_mut);
}

constexpr int FAST_RR = 8;

void launch_fast_caches(const Frame& f, Workspace& ws, hipStream_t stream, hipStream_t stream2, hipEvent_t ev,
                        hipEvent_t ev_join, hipEvent_t* timing) {
  // phase B (few long sequential rays) runs beside phase A (many short samples) on a second stream
  (void)hipEventRecord(ev, stream);
  (void)hipStreamWaitEvent(stream2, ev, 0);
  (void)hipEventRecord(timing[2], stream2);
  launch_fast_paths(f, ws, stream2, 1, f.n_path_cap);
  (void)hipEventRecord(timing[3], stream2);
  (void)hipEventRecord(ev_join, stream2);
  (void)hipEventRecord(timing[0], stream);
  hipLaunchKernelGGL(k_fast_columns, dim3(cdiv(f.wl, 256)), dim3(256), 0, stream, f, ws.colcalc);
  if (f.n_objects) {
    ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_terrain_profile<CALC, true>),
                                                          dim3(cdiv(f.n_t, PROFILE_SAMPLES_PER_BLOCK), cdiv(f.wl, 64)),
                                                          dim3(256), 0, stream, f, ws.colcalc, ws.prof, ws.plat, ws.plon));
  } else {
    ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_terrain_profile<CALC, false>),
                                                          dim3(cdiv(f.n_t, PROFILE_SAMPLES_PER_BLOCK), cdiv(f.wl, 64)),
                                                          dim3(256), 0, stream, f, ws.colcalc, ws.prof, ws.plat, ws.plon));
  }
  (void)hipEventRecord(timing[1], stream);
  (void)hipStreamWaitEvent(stream, ev_join, 0);
}

static void launch_fast_intersect_segment(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream, int i_begin,
                                          int i_end, int last) {
  dim3 grid(cdiv(f.wl, 64), cdiv(f.h, 4 * FAST_RR));
  if (f.opaque)
    hipLaunchKernelGGL((k_fast_intersect<FAST_RR, 0>), grid, dim3(256), 0, stream, f, ws.prof, ws.pelev, ws.npath,
                       ws.hit_step, out.hit_count, ws.px_steps, (unsigned long long*)ws.counters, ws.dprev, i_begin, i_end, last,
                       (uint32_t*)nullptr, (uint32_t*)nullptr, (const uint8_t*)nullptr);
  else
    hipLaunchKernelGGL((k_fast_intersect<FAST_RR, 1>), grid, dim3(256), 0, stream, f, ws.prof, ws.pelev, ws.npath,
                       ws.hit_step, out.hit_count, ws.px_steps, (unsigned long long*)ws.counters, ws.dprev, i_begin, i_end, last,
                       ws.slot_step, (uint32_t*)nullptr, (const uint8_t*)nullptr);
}

void launch_fast_intersect(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream) {
  launch_fast_intersect_segment(f, ws, out, stream, 1, f.n_path_cap, 1);
}

// Phases A, B and C of a Fast frame without scene objects.  The ray paths are one long dependent chain per row (3.5 ms at the
// headline size on a tenth of the chip) and the intersect scan only ever needs the samples integrated so far: the paths are
// integrated in FAST_SEGMENTS pieces on the second stream and the scan of piece k (main stream, after the terrain profile)
// waits for piece k alone, so it overlaps the integration of piece k + 1.  timing[4] is recorded before the first scan.
int launch_fast_pipeline(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream, hipStream_t stream2,
                         hipEvent_t ev_fork, hipEvent_t* ev_seg, hipEvent_t* ev_scan, hipEvent_t* timing) {
  const int cap = f.n_path_cap;
  const int nseg = cap >= 256 ? FAST_SEGMENTS : 1;
  int per = (cap - 1 + nseg - 1) / nseg;
  per = (per + 3) / 4 * 4; // whole chunks of the scan
  (void)hipEventRecord(ev_fork, stream);
  (void)hipStreamWaitEvent(stream2, ev_fork, 0);
  (void)hipEventRecord(timing[2], stream2);
  for (int k = 0; k < nseg; k++) {
    const int b0 = 1 + k * per, b1 = k == nseg - 1 ? cap : (b0 + per < cap ? b0 + per : cap);
    launch_fast_paths(f, ws, stream2, b0, b1);
    (void)hipEventRecord(ev_seg[k], stream2);
  }
  (void)hipEventRecord(timing[3], stream2);
  (void)hipEventRecord(timing[0], stream);
  hipLaunchKernelGGL(k_fast_columns, dim3(cdiv(f.wl, 256)), dim3(256), 0, stream, f, ws.colcalc);
  ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_terrain_profile<CALC, false>),
                                                        dim3(cdiv(f.n_t, PROFILE_SAMPLES_PER_BLOCK), cdiv(f.wl, 64)), dim3(256), 0,
                                                        stream, f, ws.colcalc, ws.prof, ws.plat, ws.plon));
  (void)hipEventRecord(timing[1], stream);
  (void)hipEventRecord(timing[4], stream);
  for (int k = 0; k < nseg; k++) {
    const int b0 = 1 + k * per, b1 = k == nseg - 1 ? cap : (b0 + per < cap ? b0 + per : cap);
    (void)hipStreamWaitEvent(stream, ev_seg[k], 0);
    (void)hipEventRecord(ev_scan[2 * k], stream); // stamped once the wait is over
    launch_fast_intersect_segment(f, ws, out, stream, b0, b1, k == nseg - 1 ? 1 : 0);
    (void)hipEventRecord(ev_scan[2 * k + 1], stream);
  }
  return nseg;
}

void launch_fast_finalize(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream) {
  ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_fast_finalize<CALC>), dim3(cdiv(f.wl, 256), f.h), dim3(256), 0,
                                                        stream, f, ws.colcalc, ws.prof, ws.pelev, ws.plen, ws.hit_step, out));
}

void launch_scan_u32(const uint32_t* in, size_t n, uint64_t* tmp, uint64_t* out, unsigned long long* total,
                     hipStream_t stream) {
  unsigned nb = cdiv(n, 256 * SCAN_ITEMS);
  hipLaunchKernelGGL(k_scan_block_sums, dim3(nb), dim3(256), 0, stream, in, n, tmp);
  hipLaunchKernelGGL(k_scan_sums, dim3(1), dim3(256), 0, stream, tmp, (size_t)nb, total);
  hipLaunchKernelGGL(k_scan_apply, dim3(nb), dim3(256), 0, stream, in, n, tmp, out);
}

void launch_scan_counts(const Frame& f, Workspace& ws, const uint32_t* hit_count, hipStream_t stream) {
  launch_scan_u32(hit_count, (size_t)f.wl * f.h, ws.scan_tmp, ws.hit_offset, (unsigned long long*)ws.counters, stream);
}

void launch_close_count(const Frame& f, Workspace& ws, hipStream_t stream) {
  size_t n = (size_t)f.n_t * f.wl;
  ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_column_candidates<CALC>), dim3(cdiv(f.wl, 64)), dim3(64), 0, stream, f,
                                                        ws.colcalc, ws.col_cand, ws.col_ncand, ws.col_lo, ws.col_hi,
                                                        (unsigned long long*)ws.counters));
  hipLaunchKernelGGL((k_close_objects<false>), dim3(cdiv(n, 256)), dim3(256), 0, stream, f, ws.plat, ws.plon, ws.col_cand,
                     ws.col_ncand, ws.ccount, (const uint64_t*)nullptr, (uint32_t*)nullptr);
  // total number of list entries -> counters[3]
  launch_scan_u32(ws.ccount, n, ws.scan_tmp, ws.coffset, (unsigned long long*)ws.counters + 2, stream);
}

void launch_close_fill(const Frame& f, Workspace& ws, hipStream_t stream) {
  size_t n = (size_t)f.n_t * f.wl;
  hipLaunchKernelGGL((k_close_objects<true>), dim3(cdiv(n, 256)), dim3(256), 0, stream, f, ws.plat, ws.plon, ws.col_cand,
                     ws.col_ncand, ws.ccount, ws.coffset, ws.clist);
}

void launch_pack_first_hits(const Frame& f, Workspace& ws, const DensePlanes& dense, const PackedHits& packed,
                            hipStream_t stream) {
  size_t n = (size_t)f.wl * f.h;
  hipLaunchKernelGGL(k_pack_first_hits, dim3(cdiv(n, 256)), dim3(256), 0, stream, f, ws.hit_offset, dense, packed);
}

void launch_multi_fill_fast(const Frame& f, Workspace& ws, uint64_t n_hits, const DensePlanes& dense,
                            const PackedHits& packed, hipStream_t stream) {
  hipLaunchKernelGGL(k_fast_gather_steps, dim3(cdiv((size_t)f.wl * f.h, 256)), dim3(256), 0, stream, f, (const uint32_t*)dense.hit_count,
                     ws.hit_offset, ws.slot_step, ws.list_step, ws.list_pixel);
  hipLaunchKernelGGL(k_fast_list, dim3(cdiv(f.wl, 256), f.h), dim3(256), 0, stream, f, ws.prof, ws.pelev, ws.npath,
                     ws.hit_offset, (const uint32_t*)dense.hit_count, ws.list_step, ws.list_pixel);
  if (n_hits)
    ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_fast_finalize_list<CALC>), dim3(cdiv(n_hits, 256)), dim3(256), 0,
                                                          stream, f, n_hits, ws.colcalc, ws.prof, ws.pelev, ws.plen,
                                                          ws.list_step, ws.list_pixel, packed));
  launch_dense_from_packed(f, ws, packed, dense, 1, stream);
}

void launch_trace_count(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream) {
  if (f.p.generator == ATMRT_GEN_RECTILINEAR) {
    launch_rect_trace_count(f, ws, out, stream);
    return;
  }
  hipLaunchKernelGGL(k_paths_transpose, dim3(f.n_path_cap, cdiv(f.h, 256)), dim3(256), 0, stream, f, ws.pelev, ws.plen, ws.npath,
                     ws.pelev_t, ws.plen_t);
  // Counting pass in three parts: (1) mark the pixels that can have a step with an object (k_fast_flag_rows); (2) the plain
  // intersect scan over every pixel — terrain crossings only, at 6 instructions per ray-step — into the tracer's slot arena;
  // (3) the general tracer (one column per wavefront, sequential over the samples) only over the marked rows, overwriting what
  // the scan left for them.  Config 5: k_fast_trace<false> 14.2 -> 2.3 ms (+ 3.1 ms of scan, 0.3 ms of marking); frame 22.4 -> 15.2 ms.
  hipLaunchKernelGGL(k_fast_flag_rows, dim3(f.wl, cdiv(f.h, 256)), dim3(256), 0, stream, f, ws.col_cand, ws.col_ncand, ws.col_lo, ws.col_hi,
                     ws.pelev_t, ws.npath, ws.traced);
  {
    dim3 grid(cdiv(f.wl, 64), cdiv(f.h, 4 * FAST_RR));
    if (f.p.terrain_alpha == 1.0)
      hipLaunchKernelGGL((k_fast_intersect<FAST_RR, 0>), grid, dim3(256), 0, stream, f, ws.prof, ws.pelev, ws.npath, ws.hit_step,
                         out.hit_count, ws.px_steps, (unsigned long long*)ws.counters, ws.dprev, 1, f.n_path_cap, 1, ws.slot_step,
                         ws.slot_packed.color_tag, (const uint8_t*)ws.traced);
    else
      hipLaunchKernelGGL((k_fast_intersect<FAST_RR, 1>), grid, dim3(256), 0, stream, f, ws.prof, ws.pelev, ws.npath, ws.hit_step,
                         out.hit_count, ws.px_steps, (unsigned long long*)ws.counters, ws.dprev, 1, f.n_path_cap, 1, ws.slot_step,
                         ws.slot_packed.color_tag, (const uint8_t*)ws.traced);
  }
  hipLaunchKernelGGL((k_fast_trace<false>), dim3(cdiv(f.wl, 4), cdiv(f.h, 64)), dim3(256), 0, stream, f, ws.prof, ws.plat, ws.plon,
                     ws.ccount, ws.coffset, ws.clist, ws.pelev_t, ws.plen_t, ws.npath, out.hit_count, (const uint64_t*)nullptr,
                     ws.slot_packed, ws.slot_step, ws.slot_pixel, ws.px_steps, (unsigned long long*)ws.counters, (double*)nullptr,
                     (const uint8_t*)ws.traced);
}

void launch_trace_fill(const Frame& f, Workspace& ws, uint64_t n_hits, const DensePlanes& dense, const PackedHits& packed,
                       hipStream_t stream) {
  if (f.p.generator == ATMRT_GEN_RECTILINEAR) {
    launch_rect_trace_fill(f, ws, n_hits, dense, packed, stream);
    return;
  }
  hipLaunchKernelGGL(k_fast_gather_trace_slots, dim3(cdiv((size_t)f.wl * f.h, 256)), dim3(256), 0, stream, f,
                     (const uint32_t*)dense.hit_count, ws.hit_offset, ws.slot_step, ws.slot_packed, ws.list_step, ws.list_pixel, packed);
  hipLaunchKernelGGL((k_fast_trace<true>), dim3(cdiv(f.wl, 4), cdiv(f.h, 64)), dim3(256), 0, stream, f, ws.prof, ws.plat, ws.plon,
                     ws.ccount, ws.coffset, ws.clist, ws.pelev_t, ws.plen_t, ws.npath, dense.hit_count, ws.hit_offset, packed,
                     ws.list_step, ws.list_pixel, (uint32_t*)nullptr, (unsigned long long*)ws.counters, ws.step_prop,
                     (const uint8_t*)nullptr);
  if (n_hits) {
    ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_fast_finalize_list<CALC>), dim3(cdiv(n_hits, 256)), dim3(256), 0,
                                                          stream, f, n_hits, ws.colcalc, ws.prof, ws.pelev, ws.plen,
                                                          ws.list_step, ws.list_pixel, packed));
  }
  launch_dense_from_packed(f, ws, packed, dense, 1, stream);
}

void launch_dense_from_packed(const Frame& f, Workspace& ws, const PackedHits& packed, const DensePlanes& dense, int fast_angles,
                              hipStream_t stream) {
  hipLaunchKernelGGL(k_dense_from_packed, dim3(cdiv(f.wl, 256), f.h), dim3(256), 0, stream, f, ws.hit_offset, packed, dense,
                     fast_angles);
}

void launch_get_elev(const Frame& f, size_t n, const double* lat, const double* lon, double* elev, uint8_t* valid,
                     hipStream_t stream) {
  if (n) hipLaunchKernelGGL(k_get_elev, dim3(cdiv(n, 256)), dim3(256), 0, stream, f, n, lat, lon, elev, valid);
}
void launch_ray_paths(const Frame& f, double h0, size_t n_angles, const double* angles_deg, int straight, double step,
                      size_t n_steps, double* x, double* h, hipStream_t stream) {
  if (n_angles)
    hipLaunchKernelGGL(k_ray_paths, dim3(cdiv(n_angles, 64)), dim3(64), 0, stream, f, h0, n_angles, angles_deg,
                       straight, step, n_steps, x, h);
}
void launch_atm_sample(const Frame& f, size_t n, const double* alt, double* t, double* p, double* nidx, double* dn,
                       hipStream_t stream) {
  if (n) hipLaunchKernelGGL(k_atm_sample, dim3(cdiv(n, 256)), dim3(256), 0, stream, f, n, alt, t, p, nidx, dn);
}
void launch_coords_at_dist(const Frame& f, double lat0, double lon0, double dir, size_t n, const double* dist,
                           double* lat, double* lon, hipStream_t stream) {
  if (n)
    hipLaunchKernelGGL(k_coords_at_dist, dim3(cdiv(n, 256)), dim3(256), 0, stream, f, lat0, lon0, dir, n, dist, lat, lon);
}

} // namespace atmrt

// atmrt_march_impl.h — Rectilinear generator on gfx950: the per-pixel march (RK4 ray + geodesic point + terrain gather per step),
// its trace-point epilogue and the general tracer for scenes with objects.  Included by four translation units
// (atmrt_march_{linear,spline}.hip, atmrt_trace_{linear,spline}.hip) that instantiate the launchers of the lean march / of the
// general tracer for atmospheres without / with Spline segments, so the heavy template variants (4 + 2 modes x 4 DirectionalCalc
// kinds x 2) compile in parallel.
#pragma once
// exp/log tables of detmath.h in LDS for this translation unit (6 KB per block): every kernel below calls stage_dm_tables() first.
// The look-ups sit on the dependent chain of each n(h) evaluation; from LDS they cost ~1/3 of an L1 hit.
#if defined(__HIP_DEVICE_COMPILE__)
#define DM_TABLES_LDS atmrt_dm_tables_lds
__shared__ double atmrt_dm_tables_lds[768];
#endif
// In this translation unit the per-object collision code and the proximity filter of a sample are called out of line: they run at
// a few per cent of the marching steps, and inlined they push the general tracer to 256 VGPRs (2 waves per SIMD; config 5:
// 559 ms inlined at 2 waves, 507 ms out of line at 3 waves, 680 ms at 4 waves with the spills that needs).
#ifndef ATMRT_OBJ_FN
#define ATMRT_OBJ_FN __attribute__((noinline))
#endif
#include <cstdlib>
#include <cstring>
#include "atmrt_device.h"

namespace atmrt {


// ---------------------------------------------------------------------------------------------
// Rectilinear generator: one ray per lane — per-step geodesic point, bilinear terrain gather
// (4 int16 posts = 8 B), sign test, RK4 step (rectilinear.rs:161-185 driving utils.rs:201-289).
// MODE 0: opaque, write the dense first hit.  MODE 1: count.  MODE 2: write packed trace points.
// MODE 3: scenes with objects — count like MODE 1 (slots pixel-indexed, in the general tracer's arena), but give up on a ray
// (hit_count = OBJECT_RAY) at the first step that can involve an object; those rays are traced by k_rect_trace afterwards.
// ---------------------------------------------------------------------------------------------
// The march only records WHERE the ray crossed the terrain (step index + ray elevation and path
// length at the two bracketing samples); k_rect_finalize rebuilds the geodesic points, the four
// finite-difference terrain lookups per sample and the interpolation.  Keeping the hit epilogue out
// of the march keeps the RK4 loop at ~135 VGPRs without scratch.

// Wavefronts per SIMD.  Rounds 1-3 ran 4 (<= 128 VGPRs, no scratch): the loop then was saturated by its own issue slots (VALU 99.7 %
// busy) and a fifth wavefront bought 1.5 % for 40 x the algorithmic stores in scratch traffic.  With round 4's evaluation order
//  a ray-step is 752 lane-instructions instead of 1,154 and the four wavefronts also wait on their own dependency
// chains (busy 91.7 %): 5 per SIMD (96 VGPRs, 116 B/lane of scratch for state that is touched once per ray) measure 173.3 against
// 179.3 ms for the headline frame, 6 per SIMD (80 VGPRs) 175.9.  History: 402 ms (3 waves) / 366 (4) / 419 (5) in round 1.
#ifndef ATMRT_MARCH_WAVES
#define ATMRT_MARCH_WAVES 5
#endif
// Scenes with objects (MODE 3).  The general tracer pays for its generality at every step (candidate lists, step lists,
// collision geometry: 3 waves per SIMD), yet a step can only involve an object while the ray is inside the distance interval of a
// candidate object AND its segment enters that object's height band — for most rays never.  So the lean march runs first, with one
// candidate list PER WAVEFRONT in LDS: the wavefront's 64 rays are adjacent pixels of one row, they advance through the same
// distances x (x = 0 + step + ... is the same sum for every ray), and the list holds, for every object that is a candidate of any
// of its rays, the union of those rays' intervals [lo, hi] and the object's height band.  The per-step test is a wave-uniform
// scan of that list, executed only between the list's next interval start and the end of the intervals it is inside.  The union
// is a superset of every ray's own candidates (same candidate_interval as ray_candidates), the band test is the tracer's own
// (object_out_of_band), so a ray that is never flagged has provably no step with an object: its terrain-only result is final.
constexpr uint32_t OBJECT_RAY = 0xffffffffu; // hit_count of a ray left to k_rect_trace

// Wave priority against the drain at the end of a launch.  The SIMD issues oldest-wave-first, so the four wavefronts that share a
// SIMD in the LAST resident set do not finish together: the favoured one runs at its dependency-chain speed, and the last one
// ends alone on a SIMD it cannot fill (a launch of uniform 2000-step rays takes 4.02 ms per workgroup-per-CU + 3.5 ms: one resident
// set 19.5 ms, not 16.1).  A wavefront therefore starts at priority 3 and drops one level every DRAIN_PRIO_BAND steps: the
// wavefronts of a SIMD are kept within a band of one another and the last set drains together (16.8 ms; a shard of the headline at
// 8 GPUs 41.5 -> 39.0 ms).  In a launch of many resident sets the drain is 1 % and the priorities cost about as much (the headline
// frame 269.4 -> 271.6 ms), so only launches of at most DRAIN_PRIO_MAX_BLOCKS workgroups (16 resident sets) use them: the kernel is
// compiled in both variants (DRAIN) and the launcher picks one by the size of its grid (ATMRT_LAUNCH_MARCH).
constexpr unsigned DRAIN_PRIO_MAX_BLOCKS = MARCH_SMALL_MAX_BLOCKS;
constexpr int DRAIN_PRIO_BAND = 512;
// Since the plain march runs 5 wavefronts per SIMD too (round 4) the small-launch variant only pays where the drain is a large part
// of the launch: measured on the tiles of the headline (profiles/r04/march_occupancy_and_slices.json), translucent terrain (MODE 1)
// plain / small at 16384 blocks 195 / 199 ms, at 8192 203 / 208, at 4096 235 / 218; scenes with objects (MODE 3, where the variant also
// does the object steps out of line) 251 / 258, 295 / 267, 384 / 274.
template <int MODE>
constexpr unsigned drain_max_blocks() {
  return MODE == 1 ? 6144u : MODE == 3 ? 12288u : DRAIN_PRIO_MAX_BLOCKS;
}
#ifndef ATMRT_MARCH_BLOCK_SMALL
#define ATMRT_MARCH_BLOCK_SMALL 256 // workgroup size of the small-launch (DRAIN) variant
#endif
// launches k_rect_march<MODE, CALC, CUBIC, DRAIN> over N rays / list entries, DRAIN by the grid size
#define ATMRT_LAUNCH_MARCH(MODE, N, STREAM, ...)                                                                                       \
  do {                                                                                                                                 \
    const unsigned blocks_ = cdiv((size_t)(N), 256);                                                                                   \
    const int override_ = march_variant_override();                                                                                    \
    if (override_ ? override_ != 1 : blocks_ <= drain_max_blocks<MODE>()) {                                                            \
      ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_rect_march<MODE, CALC, CUBIC, true>),                                    \
                                                            dim3(cdiv((size_t)(N), ATMRT_MARCH_BLOCK_SMALL)), dim3(ATMRT_MARCH_BLOCK_SMALL), 0, STREAM, \
                                                            __VA_ARGS__));                                                             \
    } else {                                                                                                                           \
      ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_rect_march<MODE, CALC, CUBIC, false>), dim3(blocks_), dim3(256), 0, STREAM, \
                                                            __VA_ARGS__));                                                             \
    }                                                                                                                                  \
  } while (0)

// The DRAIN variant also runs 5 wavefronts per SIMD (96 VGPRs, 136 B of scratch per lane) instead of 4 (127 VGPRs, none): a shard's
// long wavefronts are 1.37 resident sets of 4096 but 1.09 sets of 5120, so far fewer of them are left to run on half-empty SIMDs at
// the end (mean occupancy of a shard launch at 8 GPUs 0.83, VALU busy 0.86 with 4 per SIMD): 8 x 39.3 -> 8 x 36.2 ms, 4 x 73.5 ->
// 4 x 71.2.  (Round 4: the plain and the sliced march run 5 per SIMD too, ATMRT_MARCH_WAVES.)
#ifndef ATMRT_MARCH_WAVES_SMALL
#define ATMRT_MARCH_WAVES_SMALL 5
#endif
// a crossing beyond the slots of its pixel: one record in the arena (0.7 % of the headline's pixels at terrain_alpha 0.5 have any)
static __device__ __forceinline__ void overflow_append(const OverflowArena& ovf, unsigned long long* counters, uint32_t p, unsigned ordinal,
                                                       uint32_t step, double re0, double pl0, double re1, double pl1) {
  if (!ovf.cap) return;
  const unsigned long long k = atomicAdd(&counters[13], 1ull);
  if (k < ovf.cap) {
    ovf.pixel[k] = p;
    ovf.ordinal[k] = ovf.color_tag ? ordinal | OVERFLOW_LEAN : ordinal;
    ovf.step[k] = step;
    ovf.re0[k] = re0;
    ovf.pl0[k] = pl0;
    ovf.re1[k] = re1;
    ovf.pl1[k] = pl1;
    if (ovf.color_tag) ovf.color_tag[k] = ATMRT_COLOR_TERRAIN;
  }
}

// ---------------------------------------------------------------------------------------------
// MODE 3, the rare step.  Until round 3 a ray of the lean march that reached a step which could involve an object was abandoned
// there and traced again from x = 0 by the general tracer k_rect_trace — 11 % of config 5's rays, 38 ms of 333 for the frame and a
// launch of under two wavefronts per SIMD for a column tile (12 ms at the speed of a lone wavefront's dependency chain whatever the
// tile's size: config 5 did not strong-scale, 0.73 at 8 tiles).  Yet the steps that involve an object are a handful per such ray
// (the step must lie inside an object's distance interval AND enter its height band): 0.05 % of all steps.  So the lean march now
// keeps the ray and hands only THAT STEP to an out-of-line function that does what get_single_pixel does for a step in full
// (utils.rs:211-285): both samples' geodesic points and terrain heights, the terrain crossing, the proximity filter of the two
// samples over the wavefront's candidate objects (TerrainData::from_lat_lon, utils.rs:74-80), the collisions (frustum.rs:18-101,
// billboard.rs), the stable sort by `prop`, the emission into the tracer's slot arena / overflow arena — the general tracer's own
// building blocks (step_push, step_object_impl, step_emit), so the trace points are the tracer's to the bit.  The function is
// called under divergent control flow a few times per flagged ray; it reads the frame and the sinks through pointers to copies in
// HBM (a kernel argument cannot be addressed) and is built, like every unit that calls device functions, without interprocedural
// register allocation (Makefile CALL_EXTRA).  Only a wavefront whose candidate list overflows (more than WAVE_CAND objects) or an
// earth model without the geometric pre-filter still leaves its rays to k_rect_trace.
struct ObjectStepIO {      // one lane's step, in scratch at the call site
  DirCalc c;
  double d0, sx, re0, sh, pl0, path_length; // the two samples: stepper x, ray elevation, path length
  double diff1;            // out: ray - terrain at the newer sample (the next step's diff0)
  uint32_t pixel;
  int32_t step_index;      // of the older sample
  unsigned count;          // in / out: trace points of the pixel so far
  int32_t finish;          // out: the ray ends with this step (utils.rs:237-239, 274-285)
};
template <int CALC>
static __device__ __attribute__((noinline)) void object_step_impl(const Frame* __restrict__ fg, const ObjectStepSinks* __restrict__ sk,
                                                                  ObjectStepIO* io, const double* w_lo, const double* w_hi,
                                                                  const int* w_obj, int n_e) {
  Earth e = fg->earth;
  e.calc = CALC;
  const DirCalc& c = io->c;
  const double d0 = io->d0, sx = io->sx, re0 = io->re0, sh = io->sh, pl0 = io->pl0, path_length = io->path_length;
  double lat0, lon0, lat1, lon1;
  coords_at_dist(e, c, d0, lat0, lon0);
  coords_at_dist(e, c, sx, lat1, lon1);
  const double te0 = terrain_elev_or_zero(fg->tv, lat0, lon0), te1 = terrain_elev_or_zero(fg->tv, lat1, lon1);
  const double diff1 = re0 - te0, diff2 = sh - te1;
  StepHits hits;
  hits.n = 0;
  hits.finish = false;
  if (diff1 * diff2 < 0.0) { // utils.rs:222
    step_push(hits, diff1 / (diff1 - diff2), -1, nullptr);
    if (fg->p.terrain_alpha == 1.0) hits.finish = true;
  }
  // the objects this step tests, ascending: those close to either sample (utils.rs:241-250) whose height band the segment enters
  const Vec3 pos1 = as_cartesian(e, lat0, lon0, re0), pos2 = as_cartesian(e, lat1, lon1, sh);
  const LatLonTrig t0 = latlon_trig(e, lat0, lon0), t1 = latlon_trig(e, lat1, lon1);
  const double x_prev = sx - fg->p.simulation_step * 1.000001;
  for (int q = 0; q < n_e; q++) {
    if (w_hi[q] < x_prev || w_lo[q] > sx) continue; // the wavefront's interval of this object does not reach the step
    const int j = w_obj[q];
    const ObjectDev& o = fg->objects[j];
    if (object_out_of_band(o, re0, sh)) continue;
    if (!(object_is_close(e, o, t0) || object_is_close(e, o, t1))) continue;
    step_object_impl(hits, fg->objects, fg->textures, j, pos1, pos2);
  }
  // emission: the counting pass of the general tracer (k_rect_trace<false>)
  const unsigned count = io->count;
  const uint32_t p = io->pixel;
  if (hits.n > STEP_CANDIDATES) atomicAdd(&sk->counters[6], 1ull); // the fill pass will need Workspace::step_prop
  if (hits.n && count + (unsigned)hits.n <= (unsigned)RECT_SLOTS) {
    uint64_t k = (uint64_t)p * RECT_SLOTS + count;
    const uint64_t k0 = k;
    step_emit(hits, sk->slot_packed, sk->slot_step, sk->slot_pixel, k, p, io->step_index, lat0, lon0, re0, d0, pl0, lat1, lon1, sh, sx, path_length);
    for (uint64_t q = k0; q < k; q++) { // terrain points: what k_rect_finalize_list needs
      sk->slots.re0[q] = re0;
      sk->slots.pl0[q] = pl0;
      sk->slots.re1[q] = sh;
      sk->slots.pl1[q] = path_length;
    }
  } else if (hits.n && hits.n <= STEP_CANDIDATES && sk->ovf.cap) { // beyond the slots: the step's points into the overflow arena
    const unsigned long long base = atomicAdd(&sk->counters[13], (unsigned long long)hits.n);
    if (base + (unsigned long long)hits.n <= sk->ovf.cap) {
      uint64_t kk = base;
      step_emit(hits, sk->ovf_packed, sk->ovf.step, sk->ovf.pixel, kk, p, io->step_index, lat0, lon0, re0, d0, pl0, lat1, lon1, sh, sx, path_length);
      for (uint64_t q = base; q < kk; q++) {
        sk->ovf.ordinal[q] = count + (unsigned)(q - base);
        sk->ovf.re0[q] = re0;
        sk->ovf.pl0[q] = pl0;
        sk->ovf.re1[q] = sh;
        sk->ovf.pl1[q] = path_length;
      }
    }
  }
  atomicAdd(&sk->counters[14], 1ull); // statistics (atmrt_last_stats().object_steps)
  io->count = count + (unsigned)hits.n;
  io->finish = hits.finish ? 1 : 0;
  io->diff1 = diff2;
}

#ifdef ATMRT_TIMELINE
// Experiment hook (tools/measure_march_timeline.py; never defined in the product build): start / end time and steps of every
// wavefront of the last k_rect_march launch, read back through atmrt_debug_timeline.
static __device__ unsigned long long g_timeline[3 * 65536 + 4 * 32768];
static __device__ unsigned long long g_slices[4 * 262144 + 8]; // [0] = count; then {group | i0 << 32, start, end, wave | hw_id << 32}
#endif
template <int MODE, int CALC, bool CUBIC, bool DRAIN>
__global__ __launch_bounds__(DRAIN ? ATMRT_MARCH_BLOCK_SMALL : 256, DRAIN ? ATMRT_MARCH_WAVES_SMALL : ATMRT_MARCH_WAVES) void k_rect_march(Frame f, DensePlanes out, int32_t* __restrict__ hit_step,
                                                    const uint64_t* __restrict__ hit_offset, RectRec rec,
                                                    uint32_t* __restrict__ list_step, uint32_t* __restrict__ list_pixel,
                                                    unsigned long long* __restrict__ counters,
                                                    const uint32_t* __restrict__ pixel_list, uint32_t n_list, OverflowArena ovf,
                                                    const Frame* __restrict__ frame_dev, const ObjectStepSinks* __restrict__ sinks_dev) {
#ifdef ATMRT_TIMELINE
  const unsigned long long tl_t0 = wall_clock64();
#endif
  stage_dm_tables();
  constexpr bool drain_prio = DRAIN;
  if (drain_prio) __builtin_amdgcn_s_setprio(3);
  const size_t plane = (size_t)f.wl * f.h;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  // MODE 2 may be restricted to a list of pixels (those whose crossings did not fit the slots of the counting march)
  const bool live = pixel_list ? tid < n_list : tid < plane;
  const size_t p = pixel_list ? (live ? pixel_list[tid] : 0) : tid;
  unsigned long long steps = 0, lookups = 0;
  if (live) {
    const Earth e = earth_for<CALC>(f);
    const int y = (int)(p / (size_t)f.wl), x = (int)(p % (size_t)f.wl);
    const bool sph = e.spherical != 0;
    const double radius = e.shape_radius;
    const bool straight = f.p.straight_rays != 0;
    const double step = f.p.simulation_step, max_dist = f.p.frame.max_distance;
    const double skip_above = f.tv.skip_above;
    const bool opaque = f.p.terrain_alpha == 1.0;
    const double alt = *f.alt;
    double direction, elevation;
    rect_ray_params(f.p, f.ph, f.c0 + x, y, direction, elevation);
    DirCalc c;
    dircalc_new(e, f.p.position.latitude, f.p.position.longitude, dm_to_degrees(direction), c);
    Stepper s;
    stepper_init(s, sph, radius, alt, elevation);
    unsigned count = 0;
    int first = -1;
    uint64_t k = MODE == 2 ? hit_offset[p] : 0;
    bool object_ray = false; // MODE 3: this ray is left to the general tracer
    int w_n = 0;             // MODE 3: entries of the wavefront's candidate list
    double x_wake = dm_inf();
    __shared__ double w_lo[MODE == 3 ? 4 : 1][MODE == 3 ? WAVE_CAND : 1], w_hi[MODE == 3 ? 4 : 1][MODE == 3 ? WAVE_CAND : 1],
        w_vlo[MODE == 3 ? 4 : 1][MODE == 3 ? WAVE_CAND : 1], w_vhi[MODE == 3 ? 4 : 1][MODE == 3 ? WAVE_CAND : 1];
    __shared__ int w_obj[MODE == 3 ? 4 : 1][MODE == 3 ? WAVE_CAND : 1];
    const int wv = threadIdx.x >> 6;
    if (MODE == 3) {
      if (!candidates_supported<CALC>(e)) {
        object_ray = true; // no geometric pre-filter for this earth model: every ray tests every object
      } else {
        const Vec3 nrm = track_normal<CALC>(c);
        for (int j = 0; j < f.n_objects; j++) { // wave-uniform loop; the object's fields are scalar loads
          double lo = dm_inf(), hi = -dm_inf();
          double l, h;
          if (candidate_interval<CALC>(e, c, nrm, f.objects[j], true, l, h)) lo = l, hi = h;
          unsigned long long holders = __ballot(lo <= hi); // the active lanes for which the object is a candidate
          if (holders) {
            double wlo = dm_inf(), whi = -dm_inf();
            for (; holders; holders &= holders - 1) { // union of their intervals (reads active lanes only)
              const int src = __builtin_ctzll(holders);
              const double l2 = __shfl(lo, src, 64), h2 = __shfl(hi, src, 64);
              wlo = l2 < wlo ? l2 : wlo;
              whi = h2 > whi ? h2 : whi;
            }
            lo = wlo;
            hi = whi;
            if (w_n < WAVE_CAND) {
              if ((threadIdx.x & 63) == 0) {
                w_lo[wv][w_n] = lo;
                w_hi[wv][w_n] = hi;
                w_vlo[wv][w_n] = f.objects[j].vlo;
                w_vhi[wv][w_n] = f.objects[j].vhi;
                w_obj[wv][w_n] = j;
              }
            }
            w_n++;
          }
        }
        if (w_n > WAVE_CAND) object_ray = true; // the list overflowed: leave the whole wavefront to the tracer
        // lane 0 wrote the list, every lane of the wavefront reads it from here on: DS operations of one wave issue in order, but the
        // dependence has to exist for the compiler too — a wavefront-scope release / acquire pair around a wave barrier
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int q = 0; q < w_n && q < WAVE_CAND; q++) // first interval that is not behind the start
          if (w_hi[wv][q] >= 0.0) x_wake = w_lo[wv][q] < x_wake ? w_lo[wv][q] : x_wake;
      }
    }
    // first sample (PathIterator::next at the start state); the reference would panic on an empty stream
    if (!(0.0 > max_dist || alt < -1000.0) && !object_ray) {
      double lat, lon;
      coords_at_dist(e, c, 0.0, lat, lon);
      double diff0 = alt - terrain_elev_or_zero(f.tv, lat, lon);
      lookups++;
      double re0 = alt, pl0 = 0.0; // TracingState::new(.., first_path.elev, 0.0, 0.0), utils.rs:208
      double sx = 0.0, sh = alt, path_length = 0.0;
      for (int i = 1;; i++) {
        if (drain_prio && (i & (DRAIN_PRIO_BAND - 1)) == 0) { // see DRAIN_PRIO_MAX_BLOCKS
          const int band = i / DRAIN_PRIO_BAND;
          if (band == 1) __builtin_amdgcn_s_setprio(2);
          else if (band == 2) __builtin_amdgcn_s_setprio(1);
          else if (band == 3) __builtin_amdgcn_s_setprio(0);
        }
        bool tame;
        RayState st = stepper_next<CUBIC>(s, *f.atm, sph, radius, straight, step, tame);
        if (straight) tame = __all(calc_dist_in_band(*f.atm, sh) && calc_dist_in_band(*f.atm, st.h));
        path_length += calc_dist(sph, radius, sx, sh, st.x, st.h, tame, f.inv_shape_radius);
        sx = st.x;
        sh = st.h;
        if (sx > max_dist || sh < -1000.0 || !(sx <= max_dist)) break; // rectilinear.rs:178 (+ NaN guard)
        if (MODE == 3 && sx >= x_wake) { // wave-uniform: x and the list are the same for every ray of the wavefront
          // this step (the samples at x - step and x) against the list: inside an entry's interval and inside its height band ->
          // the step may involve that object: the ray goes to the tracer.  Next wake: the nearest interval start ahead, or the
          // next step while an interval is being crossed.
          const double x_prev = sx - step * 1.000001; // a little before the older sample (its x is sx - step up to rounding)
          double next = dm_inf();
          const int n_e = w_n < WAVE_CAND ? w_n : WAVE_CAND;
          bool object_step = false;
          for (int q = 0; q < n_e; q++) {
            const double lo = w_lo[wv][q], hi = w_hi[wv][q];
            if (hi < x_prev) continue;               // behind the step
            if (lo > sx) {                           // ahead of it
              next = lo < next ? lo : next;
              continue;
            }
            next = sx;                               // being crossed: look again at the next step
            const double vlo = w_vlo[wv][q], vhi = w_vhi[wv][q];
            if (!((re0 < vlo && sh < vlo) || (re0 > vhi && sh > vhi))) object_step = true; // object_out_of_band is false
          }
          x_wake = next;
          // A launch of many resident sets (the whole frame: DRAIN false) abandons the ray here and leaves it to the general tracer,
          // whose own launch is then large enough to fill the chip (config 5: 38 ms of VALU-saturated work for 11 % of the rays —
          // there is nothing to hide, and the out-of-line steps would only add their call overhead: 308 against 328 ms per frame).
          // A small launch (a column tile of a multi-GPU frame: DRAIN true) keeps the ray and does the step out of line: the
          // tracer's launch for a tile would be under two wavefronts per SIMD, 12 ms whatever the tile's size (8 tiles: 54 -> 43 ms).
          if (object_step && !DRAIN) {
            object_ray = true;
            break;
          }
          if (object_step) { // this lane's step in full, out of line (object_step_impl above); the others wait
            ObjectStepIO io;
            io.c = c;
            io.d0 = f.xs[i - 1]; // the stepper's x of the older sample: 0 + step + ... (the same additions as xs)
            io.sx = sx, io.re0 = re0, io.sh = sh, io.pl0 = pl0, io.path_length = path_length;
            io.pixel = (uint32_t)p, io.step_index = i - 1, io.count = count;
            object_step_impl<CALC>(frame_dev, sinks_dev, &io, w_lo[wv], w_hi[wv], w_obj[wv], n_e);
            count = io.count;
            lookups += 2;
            steps++;
            if (io.finish) break;
            diff0 = io.diff1;
            re0 = sh;
            pl0 = path_length;
            continue;
          }
        }
        // A sample above every post of the mosaic is above the terrain, whatever its geodesic point: ray - terrain is positive and
        // only its SIGN enters the test below (the epilogue rebuilds the bracketing samples in full), so the geodesic point and
        // the lookup are skipped and any positive number stands for the difference.  Wavefronts are 64 columns of one row: sky
        // rows leave the terrain's height range together.  NaN heights take the full path.
        double diff1 = 1.0;
        if (!(sh > skip_above)) {
          coords_at_dist(e, c, sx, lat, lon);
          diff1 = sh - terrain_elev_or_zero(f.tv, lat, lon);
          lookups++;
        }
        steps++;
        if (diff0 * diff1 < 0.0) { // utils.rs:222
          if (MODE == 0) {
            first = i - 1; // the record is stored after the loop: opaque terrain ends the march here, the four values stay as they are
          } else if (MODE == 1) {
            if (count < (unsigned)RECT_SLOTS) { // slot arrays come in through list_step / rec, slot-major
              const size_t q = (size_t)count * plane + p;
              list_step[q] = (uint32_t)(i - 1);
              rec.re0[q] = re0;
              rec.pl0[q] = pl0;
              rec.re1[q] = sh;
              rec.pl1[q] = path_length;
            } else {
              overflow_append(ovf, counters, (uint32_t)p, count, (uint32_t)(i - 1), re0, pl0, sh, path_length);
            }
          } else if (MODE == 2) {
            list_step[k] = (uint32_t)(i - 1);
            list_pixel[k] = (uint32_t)p;
            rec.re0[k] = re0;
            rec.pl0[k] = pl0;
            rec.re1[k] = sh;
            rec.pl1[k] = path_length;
            k++;
          } else if (MODE == 3) {
            if (count < (unsigned)RECT_SLOTS) { // the general tracer's slot arena, pixel-indexed; list_pixel carries its color_tag array
              const size_t q = p * RECT_SLOTS + count;
              list_step[q] = (uint32_t)(i - 1);
              list_pixel[q] = ATMRT_COLOR_TERRAIN;
              rec.re0[q] = re0;
              rec.pl0[q] = pl0;
              rec.re1[q] = sh;
              rec.pl1[q] = path_length;
            } else {
              overflow_append(ovf, counters, (uint32_t)p, count, (uint32_t)(i - 1), re0, pl0, sh, path_length);
            }
          }
          count++;
          if (opaque) break; // utils.rs:237-239, 283-285
        }
        diff0 = diff1;
        re0 = sh;
        pl0 = path_length;
      }
      if (MODE == 0 && first >= 0) { // ray elevation and path length at the two samples that bracket the crossing
        rec.re0[p] = re0;
        rec.pl0[p] = pl0;
        rec.re1[p] = sh;
        rec.pl1[p] = path_length;
      }
    }
    if (MODE == 3) hit_step[p] = object_ray ? 1 : 0; // voids the ray's overflow records (k_rect_scatter_trace_overflow)
    if (MODE == 3 && object_ray) { // nothing of this ray counts: k_rect_trace starts it again
      out.hit_count[p] = OBJECT_RAY;
      steps = 0;
      lookups = 0;
    } else if (MODE != 2) {
      out.azimuth[p] = dm_to_degrees(direction); // not wrapped, rectilinear.rs:110-113
      out.elevation_angle[p] = dm_to_degrees(elevation);
      out.hit_count[p] = count;
    }
    if (MODE == 0) hit_step[p] = first;
    if ((MODE == 1 || (MODE == 3 && !object_ray)) && count > (unsigned)RECT_SLOTS) atomicAdd(&counters[3], 1ull);
  }
  if (MODE != 2) {
    steps = wave_sum(steps);
    lookups = wave_sum(lookups);
    if ((threadIdx.x & 63) == 0 && steps) {
      atomicAdd(&counters[0], steps);
      atomicAdd(&counters[10], lookups);
    }
  }
#ifdef ATMRT_TIMELINE
  if (MODE == 0 && (threadIdx.x & 63) == 0) {
    const size_t w = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    if (w < 65536) {
      g_timeline[3 * w] = tl_t0;
      g_timeline[3 * w + 1] = wall_clock64();
      // steps | HW_ID[15:0] (wave, simd, pipe, cu, sh, se) << 24 | XCC_ID << 40
      g_timeline[3 * w + 2] = steps | ((unsigned long long)(__builtin_amdgcn_s_getreg((15 << 11) | 4) & 0xffffu) << 24) |
                              ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu) << 40);
    }
  }
#endif
}

// ---------------------------------------------------------------------------------------------
// Time-sliced march (terrain only: MODE 0 opaque, MODE 1 translucent with the crossings counted into slots) for launches of a few
// resident sets: the column tiles of a multi-GPU frame.
//
// Why.  A tile of the headline at 8 GPUs is 16384 wavefronts of which 7688 march all 2000 steps: 1.5 resident sets of 5120.  The
// hardware hands the second set to whichever CUs drain first (the wavefront timeline of such a launch, tools/
// measure_march_timeline.py: 132 CUs take five more workgroups each, 124 take none and sit idle for the last 5 ms of 34), and no
// order of the grid changes that.  So the unit of work is made smaller than a ray: a SLICE of a group of 64 consecutive pixels.
// k_rect_march_first (an ordinary grid) marches the first `slice` steps of every ray and leaves, for the rays still marching, their
// state in HBM and their group in a FIFO; k_rect_march_cont serves the FIFO with one single-wavefront workgroup per entry: read the
// group's state, march `slice` steps, write it back, append the group again — or count it finished.  All groups advance together,
// the long ones end in the same round, and the launch drains in one slice instead of one ray.
// The state of a ray between two slices: x, a, b, the last sample's height, path length and ray-minus-terrain difference, the
// elevation angle (straight rays), step number, layer hint, crossings so far (MODE 1) — 64 B — and, written once, its geodesic
// calculator (128 B).
//
// FIFO: ctl[0] = entries claimed by readers, ctl[1] = entries written, ctl[2] = groups finished.  Entries are never reused (capacity:
// groups x slices a ray can need) and each has exactly one reader, the workgroup whose claim returned its index.  A reader whose
// entry is still empty waits for it — the writer is a wavefront that is marching, never one that waits — and gives up when every
// group has finished (its index is then beyond the last entry that will ever be written).
struct SliceState {
  double *x, *a, *b, *sh, *pl, *diff0, *ang; // [plane]
  int32_t *step, *hint;                      // [plane]; step < 0: the ray has finished
  uint32_t* count;                           // [plane] MODE 1: crossings so far
  DirCalc* calc;                             // [plane] the ray's geodesic calculator
  uint32_t* queue;                           // [cap], 0xffffffff = not written yet
  unsigned long long* ctl;                   // [4]
  uint32_t cap;
  char* glist;                               // MODE 3: [n_groups] candidate lists of SLICE_GROUP_LIST_BYTES each, or null
};
// a group's candidate list in HBM (MODE 3): the arrays of one record
struct GroupList {
  double *lo, *hi, *vlo, *vhi;
  int32_t* obj;
  double* n_and_wake; // [0] number of entries (as a double), [1] x_wake
};
static __device__ __forceinline__ GroupList group_list(char* base, uint32_t group) {
  char* q = base + (size_t)group * SLICE_GROUP_LIST_BYTES;
  GroupList g;
  g.n_and_wake = (double*)q;
  g.lo = g.n_and_wake + 2;
  g.hi = g.lo + WAVE_CAND;
  g.vlo = g.hi + WAVE_CAND;
  g.vhi = g.vlo + WAVE_CAND;
  g.obj = (int32_t*)(g.vhi + WAVE_CAND);
  return g;
}
// what a slice of a scene with objects needs besides the ray: the wavefront's list (LDS), its wake distance, and where the
// out-of-line object step finds the frame and the tracer's arenas
struct SliceObjects {
  const double *w_lo, *w_hi, *w_vlo, *w_vhi;
  const int* w_obj;
  int n_e;
  const Frame* frame_dev;
  const ObjectStepSinks* sinks_dev;
};
constexpr uint32_t SLICE_EMPTY = 0xffffffffu, SLICE_EXIT = 0xfffffffeu;
#ifndef ATMRT_SLICE_WAVES
#define ATMRT_SLICE_WAVES 5 // per SIMD, like the plain march (round 4: 2 x 93.3 -> 90.0 ms, 4 x 47.8 -> 46.2, 8 x 24.24 -> 24.01; 6: slower)
#endif

#ifdef ATMRT_TIMELINE
static __device__ __forceinline__ int timeline_wave_max(int v) {
  for (int o = 32; o; o >>= 1) {
    const int u = __shfl_xor(v, o, 64);
    v = u > v ? u : v;
  }
  return v;
}
#endif

// What a slice leaves behind besides the state: MODE 0 the first crossing, MODE 1 the slots of the counting march (k_rect_march<1>)
struct SliceSinks {
  int32_t* hit_step;   // MODE 0; MODE 3: 1 = the ray is left to the general tracer
  RectRec rec;         // MODE 0: [plane]; MODE 1: slot-major [RECT_SLOTS][plane]; MODE 3: pixel-major [plane][RECT_SLOTS] (the tracer's arena)
  uint32_t* slot_step; // MODE 1, MODE 3
  OverflowArena ovf;   // MODE 1, MODE 3: crossings beyond the slots
  uint32_t* slot_tag;  // MODE 3: the slot arena's colour tags (the march writes TERRAIN)
};

// one slice of the march of a ray whose state is in registers; returns true when the ray is still marching after step i0 + slice
template <int MODE, int CALC, bool CUBIC>
static __device__ __forceinline__ bool march_slice(const Frame& f, const Earth& e, const DirCalc& c, Stepper& s, double& sh,
                                                   double& path_length, double& diff0, double& re0, double& pl0, int i0, int slice,
                                                   int& first, unsigned& count, const SliceSinks& sinks, unsigned long long* counters, size_t p,
                                                   size_t plane, unsigned long long& steps, unsigned long long& lookups,
                                                   const SliceObjects& so, double& x_wake) {
  const bool sph = e.spherical != 0;
  const double radius = e.shape_radius;
  const bool straight = f.p.straight_rays != 0;
  const double step = f.p.simulation_step, max_dist = f.p.frame.max_distance;
  const double skip_above = f.tv.skip_above;
  double sx = s.x, lat, lon;
  const int i_end = i0 + slice;
  for (int i = i0 + 1;; i++) {
    bool tame;
    RayState nx = stepper_next<CUBIC>(s, *f.atm, sph, radius, straight, step, tame);
    if (straight) tame = __all(calc_dist_in_band(*f.atm, sh) && calc_dist_in_band(*f.atm, nx.h));
    path_length += calc_dist(sph, radius, sx, sh, nx.x, nx.h, tame, f.inv_shape_radius);
    sx = nx.x;
    sh = nx.h;
    if (sx > max_dist || sh < -1000.0 || !(sx <= max_dist)) return false; // rectilinear.rs:178 (+ NaN guard)
    if (MODE == 3 && sx >= x_wake) { // wave-uniform, as in k_rect_march<3>: this step against the group's candidate list
      const double x_prev = sx - step * 1.000001;
      double next = dm_inf();
      bool object_step = false;
      for (int q = 0; q < so.n_e; q++) {
        const double lo = so.w_lo[q], hi = so.w_hi[q];
        if (hi < x_prev) continue;
        if (lo > sx) {
          next = lo < next ? lo : next;
          continue;
        }
        next = sx;
        const double vlo = so.w_vlo[q], vhi = so.w_vhi[q];
        if (!((re0 < vlo && sh < vlo) || (re0 > vhi && sh > vhi))) object_step = true;
      }
      x_wake = next;
      if (object_step) { // this lane's step in full, out of line (object_step_impl)
        ObjectStepIO io;
        io.c = c;
        io.d0 = f.xs[i - 1];
        io.sx = sx, io.re0 = re0, io.sh = sh, io.pl0 = pl0, io.path_length = path_length;
        io.pixel = (uint32_t)p, io.step_index = i - 1, io.count = count;
        object_step_impl<CALC>(so.frame_dev, so.sinks_dev, &io, so.w_lo, so.w_hi, so.w_obj, so.n_e);
        count = io.count;
        lookups += 2;
        steps++;
        if (io.finish) return false;
        diff0 = io.diff1;
        re0 = sh;
        pl0 = path_length;
        if (i == i_end) return true;
        continue;
      }
    }
    double diff1 = 1.0; // above every post of the mosaic: see k_rect_march
    if (!(sh > skip_above)) {
      coords_at_dist(e, c, sx, lat, lon);
      diff1 = sh - terrain_elev_or_zero(f.tv, lat, lon);
      lookups++;
    }
    steps++;
    if (diff0 * diff1 < 0.0) { // utils.rs:222
      if (MODE == 0) { // opaque terrain ends the march here (utils.rs:237-239)
        first = i - 1;
        return false;
      }
      if (count < (unsigned)RECT_SLOTS) { // as k_rect_march<1> / <3>
        const size_t q = MODE == 3 ? p * RECT_SLOTS + count : (size_t)count * plane + p;
        sinks.slot_step[q] = (uint32_t)(i - 1);
        if (MODE == 3) sinks.slot_tag[q] = ATMRT_COLOR_TERRAIN;
        sinks.rec.re0[q] = re0;
        sinks.rec.pl0[q] = pl0;
        sinks.rec.re1[q] = sh;
        sinks.rec.pl1[q] = path_length;
      } else {
        overflow_append(sinks.ovf, counters, (uint32_t)p, count, (uint32_t)(i - 1), re0, pl0, sh, path_length);
      }
      count++;
      if (MODE == 3 && f.p.terrain_alpha == 1.0) return false; // opaque terrain ends the ray (utils.rs:237-239)
    }
    diff0 = diff1;
    re0 = sh;
    pl0 = path_length;
    if (i == i_end) return true;
  }
}

// the ray has ended: its per-pixel results (azimuth and elevation angle were written by the first slice)
template <int MODE>
static __device__ __forceinline__ void slice_finish(const DensePlanes& out, const SliceSinks& sinks, unsigned long long* counters, size_t p,
                                                    int first, unsigned count, double re0, double pl0, double sh, double path_length) {
  if (MODE == 0) {
    if (first >= 0) { // ray elevation and path length at the two samples that bracket the crossing
      sinks.rec.re0[p] = re0;
      sinks.rec.pl0[p] = pl0;
      sinks.rec.re1[p] = sh;
      sinks.rec.pl1[p] = path_length;
    }
    out.hit_count[p] = first >= 0 ? 1u : 0u;
    sinks.hit_step[p] = first;
  } else {
    out.hit_count[p] = count;
    if (MODE == 3) sinks.hit_step[p] = 0; // the ray stayed with the march: its overflow records count (k_rect_scatter_trace_overflow)
    if (count > (unsigned)RECT_SLOTS) atomicAdd(&counters[3], 1ull);
  }
}

// wave-uniform end of a group's slice: announce the group again, or count it as finished
static __device__ __forceinline__ void slice_requeue(const SliceState& st, uint32_t group, bool alive) {
  const bool again = __any(alive);
  if (again) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); // state before the entry that announces it
  if ((threadIdx.x & 63) == 0) {
    if (again) {
      const unsigned long long t = atomicAdd(&st.ctl[1], 1ull);
      if (t < st.cap) __hip_atomic_store(st.queue + t, group, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else {
      atomicAdd(&st.ctl[2], 1ull);
    }
  }
}

// slice 0 of every group: an ordinary grid (it is short: the launch drains in a fraction of a slice)
template <int MODE, int CALC, bool CUBIC>
__global__ __launch_bounds__(256, ATMRT_SLICE_WAVES) void k_rect_march_first(Frame f, DensePlanes out, SliceSinks sinks,
                                                                             unsigned long long* __restrict__ counters, SliceState st,
                                                                             uint32_t n_groups, int slice, const Frame* __restrict__ frame_dev,
                                                                             const ObjectStepSinks* __restrict__ sinks_dev) {
#ifdef ATMRT_TIMELINE
  const unsigned long long tl_t0 = wall_clock64();
#endif
  stage_dm_tables();
  const size_t plane = (size_t)f.wl * f.h;
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  unsigned long long steps = 0, lookups = 0;
  bool alive = false;
  if (p < plane) {
    const Earth e = earth_for<CALC>(f);
    const int y = (int)(p / (size_t)f.wl), x = (int)(p % (size_t)f.wl);
    const double alt = *f.alt;
    double direction, elevation;
    rect_ray_params(f.p, f.ph, f.c0 + x, y, direction, elevation);
    DirCalc c;
    dircalc_new(e, f.p.position.latitude, f.p.position.longitude, dm_to_degrees(direction), c);
    Stepper s;
    stepper_init(s, e.spherical != 0, e.shape_radius, alt, elevation);
    out.azimuth[p] = dm_to_degrees(direction); // not wrapped, rectilinear.rs:110-113
    out.elevation_angle[p] = dm_to_degrees(elevation);
    int first = -1;
    unsigned count = 0;
    double sh = alt, path_length = 0.0, re0 = alt, pl0 = 0.0, diff0 = 0.0;
    // MODE 3: the wavefront's candidate list, built as k_rect_march<3> builds it (LDS) and kept in HBM for the group's later slices
    __shared__ double w_lo[MODE == 3 ? 4 : 1][MODE == 3 ? WAVE_CAND : 1], w_hi[MODE == 3 ? 4 : 1][MODE == 3 ? WAVE_CAND : 1],
        w_vlo[MODE == 3 ? 4 : 1][MODE == 3 ? WAVE_CAND : 1], w_vhi[MODE == 3 ? 4 : 1][MODE == 3 ? WAVE_CAND : 1];
    __shared__ int w_obj[MODE == 3 ? 4 : 1][MODE == 3 ? WAVE_CAND : 1];
    const int wv = threadIdx.x >> 6;
    int w_n = 0;
    double x_wake = dm_inf();
    bool object_ray = false; // MODE 3: the list overflowed (or the earth model has no pre-filter): the wavefront's rays go to the tracer
    if (MODE == 3) {
      if (!candidates_supported<CALC>(e)) {
        object_ray = true;
      } else {
        const Vec3 nrm = track_normal<CALC>(c);
        for (int j = 0; j < f.n_objects; j++) { // wave-uniform loop; the object's fields are scalar loads
          double lo = dm_inf(), hi = -dm_inf();
          double l, h;
          if (candidate_interval<CALC>(e, c, nrm, f.objects[j], true, l, h)) lo = l, hi = h;
          unsigned long long holders = __ballot(lo <= hi);
          if (holders) {
            double wlo = dm_inf(), whi = -dm_inf();
            for (; holders; holders &= holders - 1) {
              const int src = __builtin_ctzll(holders);
              const double l2 = __shfl(lo, src, 64), h2 = __shfl(hi, src, 64);
              wlo = l2 < wlo ? l2 : wlo;
              whi = h2 > whi ? h2 : whi;
            }
            if (w_n < WAVE_CAND && (threadIdx.x & 63) == 0) {
              w_lo[wv][w_n] = wlo;
              w_hi[wv][w_n] = whi;
              w_vlo[wv][w_n] = f.objects[j].vlo;
              w_vhi[wv][w_n] = f.objects[j].vhi;
              w_obj[wv][w_n] = j;
            }
            w_n++;
          }
        }
        if (w_n > WAVE_CAND) object_ray = true;
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        for (int q = 0; q < w_n && q < WAVE_CAND; q++)
          if (w_hi[wv][q] >= 0.0) x_wake = w_lo[wv][q] < x_wake ? w_lo[wv][q] : x_wake;
      }
    }
    const int n_e = w_n < WAVE_CAND ? w_n : WAVE_CAND;
    const SliceObjects so{w_lo[wv], w_hi[wv], w_vlo[wv], w_vhi[wv], w_obj[wv], n_e, frame_dev, sinks_dev};
    if (!(0.0 > f.p.frame.max_distance || alt < -1000.0) && !object_ray) { // the reference would panic on an empty stream
      double lat, lon;
      coords_at_dist(e, c, 0.0, lat, lon);
      diff0 = alt - terrain_elev_or_zero(f.tv, lat, lon);
      lookups++;
      alive = march_slice<MODE, CALC, CUBIC>(f, e, c, s, sh, path_length, diff0, re0, pl0, 0, slice, first, count, sinks, counters, p, plane,
                                             steps, lookups, so, x_wake);
    }
    if (MODE == 3 && st.glist && __any(alive)) { // the group marches on: its list and wake distance for the slices to come
      const GroupList gl = group_list(st.glist, (uint32_t)(p >> 6));
      const int lane = threadIdx.x & 63;
      for (int q = lane; q < n_e; q += 64) {
        gl.lo[q] = w_lo[wv][q];
        gl.hi[q] = w_hi[wv][q];
        gl.vlo[q] = w_vlo[wv][q];
        gl.vhi[q] = w_vhi[wv][q];
        gl.obj[q] = w_obj[wv][q];
      }
      const unsigned long long still = __ballot(alive); // (a lane that ended inside the slice holds an older wake distance)
      if (lane == __builtin_ctzll(still)) {
        gl.n_and_wake[0] = (double)n_e;
        gl.n_and_wake[1] = x_wake;
      }
    }
    if (MODE == 3 && object_ray) { // left to the general tracer (k_collect_object_rays, k_rect_trace)
      st.step[p] = -1;
      out.hit_count[p] = OBJECT_RAY;
      sinks.hit_step[p] = 1;
      steps = 0;
      lookups = 0;
    } else if (alive) {
      st.x[p] = s.x;
      st.a[p] = s.a;
      st.b[p] = s.b;
      st.hint[p] = s.hint;
      st.ang[p] = s.ang;
      st.sh[p] = sh;
      st.pl[p] = path_length;
      st.diff0[p] = diff0;
      st.step[p] = slice;
      if (MODE == 1 || MODE == 3) st.count[p] = count;
      st.calc[p] = c;
    } else {
      st.step[p] = -1;
      slice_finish<MODE>(out, sinks, counters, p, first, count, re0, pl0, sh, path_length);
    }
  }
  if ((p >> 6) < n_groups) slice_requeue(st, (uint32_t)(p >> 6), alive); // (the last block may hold wavefronts past the last group)
  steps = wave_sum(steps);
  lookups = wave_sum(lookups);
#ifdef ATMRT_TIMELINE
  if ((threadIdx.x & 63) == 0 && (p >> 6) < 32768) {
    g_timeline[3 * (p >> 6)] = tl_t0;
    g_timeline[3 * (p >> 6) + 1] = wall_clock64();
    g_timeline[3 * (p >> 6) + 2] = steps;
  }
#endif
  if ((threadIdx.x & 63) == 0 && steps) {
    atomicAdd(&counters[0], steps);
    atomicAdd(&counters[10], lookups);
  }
}

// the later slices: one single-wavefront workgroup per queue entry.  NOT a persistent grid: the SIMD issues oldest-wavefront-first,
// and persistent wavefronts keep their age order for the whole launch — the oldest of a SIMD ran its slices at the speed of its
// dependency chain (0.57 ms), the youngest in 5 ms (measured, tools/measure_slice_timeline.py), groups that met the slow ones fell
// five rounds behind and were a 4.5 ms tail; s_setprio per quarter slice did not change that.  A wavefront that lives for one slice
// starts youngest and ends oldest: every slice sees every rank and the groups stay in step.
template <int MODE, int CALC, bool CUBIC>
__global__ __launch_bounds__(64, ATMRT_SLICE_WAVES) void k_rect_march_cont(Frame f, DensePlanes out, SliceSinks sinks,
                                                                           unsigned long long* __restrict__ counters, SliceState st,
                                                                           uint32_t n_groups, int slice, const Frame* __restrict__ frame_dev,
                                                                           const ObjectStepSinks* __restrict__ sinks_dev) {
  const size_t plane = (size_t)f.wl * f.h;
  const int lane = threadIdx.x;
  uint32_t item = 0;
  if (lane == 0) {
    const unsigned long long i = atomicAdd(&st.ctl[0], 1ull);
    if (i >= st.cap) {
      item = SLICE_EXIT; // more readers than entries: every entry has its reader already
    } else {
      const uint32_t* entry = st.queue + i;
      // An empty entry means the FIFO has run dry: the launch is in its last round and the groups still marching are held by
      // other wavefronts.  Poll rarely (every few microseconds, backing off to ~50): thousands of idle wavefronts polling two
      // addresses at full rate queue up on one memory channel, in front of the state traffic of the wavefronts still working.
      for (int backoff = 1;; backoff = backoff < 16 ? backoff * 2 : 16) {
        item = __hip_atomic_load(entry, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (item != SLICE_EMPTY) break;
        if (__hip_atomic_load(&st.ctl[2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= n_groups) {
          item = SLICE_EXIT;
          break;
        }
        __builtin_amdgcn_s_setprio(0);
        for (int k = 0; k < backoff; k++) __builtin_amdgcn_s_sleep(127); // 127 x 64 cycles = 3.4 us
      }
    }
  }
  item = (uint32_t)__shfl((int)item, 0, 64);
  if (item == SLICE_EXIT) return;
  // the group's state, written by the wavefront that announced it — on any XCD: agent scope it has to be (with workgroup-scope
  // fences the step counts of a frame change: stale state out of another XCD's L2)
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  stage_dm_tables();
#ifdef ATMRT_TIMELINE
  const unsigned long long tl_slice_t0 = wall_clock64();
#endif
  unsigned long long steps = 0, lookups = 0;
  const size_t p = (size_t)item * 64 + lane;
  bool alive = false;
  const int32_t i0 = p < plane ? st.step[p] : -1;
  // MODE 3: the group's candidate list back into LDS (what its first slice built)
  __shared__ double w_lo[MODE == 3 ? WAVE_CAND : 1], w_hi[MODE == 3 ? WAVE_CAND : 1], w_vlo[MODE == 3 ? WAVE_CAND : 1], w_vhi[MODE == 3 ? WAVE_CAND : 1];
  __shared__ int w_obj[MODE == 3 ? WAVE_CAND : 1];
  int n_e = 0;
  double x_wake = dm_inf();
  if (MODE == 3) {
    const GroupList gl = group_list(st.glist, item);
    n_e = (int)gl.n_and_wake[0];
    x_wake = gl.n_and_wake[1];
    for (int q = lane; q < n_e; q += 64) {
      w_lo[q] = gl.lo[q];
      w_hi[q] = gl.hi[q];
      w_vlo[q] = gl.vlo[q];
      w_vhi[q] = gl.vhi[q];
      w_obj[q] = gl.obj[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  }
  const SliceObjects so{w_lo, w_hi, w_vlo, w_vhi, w_obj, n_e, frame_dev, sinks_dev};
  if (i0 >= 0) {
    const Earth e = earth_for<CALC>(f);
    Stepper s;
    s.x = st.x[p];
    s.a = st.a[p];
    s.b = st.b[p];
    s.hint = st.hint[p];
    s.h0 = *f.alt;
    s.ang = st.ang[p];
    double sh = st.sh[p], path_length = st.pl[p], diff0 = st.diff0[p];
    const DirCalc c = st.calc[p];
    double re0 = sh, pl0 = path_length;
    int first = -1;
    unsigned count = MODE == 1 || MODE == 3 ? st.count[p] : 0u;
    alive = march_slice<MODE, CALC, CUBIC>(f, e, c, s, sh, path_length, diff0, re0, pl0, i0, slice, first, count, sinks, counters, p, plane,
                                           steps, lookups, so, x_wake);
    if (alive) {
      st.x[p] = s.x;
      st.a[p] = s.a;
      st.b[p] = s.b;
      st.hint[p] = s.hint;
      st.sh[p] = sh;
      st.pl[p] = path_length;
      st.diff0[p] = diff0;
      st.step[p] = i0 + slice;
      if (MODE == 1 || MODE == 3) st.count[p] = count;
    } else {
      st.step[p] = -1;
      slice_finish<MODE>(out, sinks, counters, p, first, count, re0, pl0, sh, path_length);
    }
  }
  if (MODE == 3) { // the group's next wake distance, from a lane that is still marching (the value is the same in all of them)
    const unsigned long long still = __ballot(alive); // (a lane that ended inside the slice holds an older one)
    if (still && lane == __builtin_ctzll(still)) group_list(st.glist, item).n_and_wake[1] = x_wake;
  }
  slice_requeue(st, item, alive);
#ifdef ATMRT_TIMELINE
  {
    const int i0max = timeline_wave_max(i0);
    const bool tl_again = __any(alive);
    if (lane == 0) {
      const unsigned long long k = atomicAdd(&g_slices[0], 1ull);
      if (k < 262144) {
        unsigned long long* r = g_slices + 8 + 4 * k;
        r[0] = item | ((unsigned long long)(unsigned)i0max << 32);
        r[1] = tl_slice_t0;
        r[2] = wall_clock64();
        r[3] = blockIdx.x | ((unsigned long long)(__builtin_amdgcn_s_getreg((15 << 11) | 4) & 0xffffu) << 32) |
               ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | 20) & 0xfu) << 48);
      }
      if (item < 32768 && !tl_again) g_timeline[3 * item + 2] = wall_clock64(); // the group's last slice ended
    }
  }
#endif
  steps = wave_sum(steps);
  lookups = wave_sum(lookups);
  if (lane == 0 && steps) {
    atomicAdd(&counters[0], steps);
    atomicAdd(&counters[10], lookups);
  }
}

// every group must have finished: anything else is reported through counters[12] and fails the frame (atmrt_api.hip)
static __global__ void k_slice_check(const unsigned long long* __restrict__ ctl, uint32_t n_groups, unsigned long long* __restrict__ counters) {
  counters[12] = ctl[2] == n_groups ? 0ull : 1ull + (n_groups > ctl[2] ? n_groups - ctl[2] : ctl[2] - n_groups);
}

// TracePoint of a recorded crossing of pixel (x, y) at step s
template <int CALC>
static __device__ __forceinline__ TracePointDev rect_hit(const Frame& f, int x, int y, int s, double re0, double pl0,
                                                         double re1, double pl1) {
  const Earth e = earth_for<CALC>(f);
  double direction, elevation;
  rect_ray_params(f.p, f.ph, f.c0 + x, y, direction, elevation);
  DirCalc c;
  dircalc_new(e, f.p.position.latitude, f.p.position.longitude, dm_to_degrees(direction), c);
  double d0 = f.xs[s], d1 = f.xs[s + 1]; // the stepper's x: 0 + step + ... (same additions as xs)
  double lat0, lon0, lat1, lon1;
  coords_at_dist(e, c, d0, lat0, lon0);
  coords_at_dist(e, c, d1, lat1, lon1);
  double te0 = terrain_elev_or_zero(f.tv, lat0, lon0);
  double te1 = terrain_elev_or_zero(f.tv, lat1, lon1);
  return terrain_trace_point(f, e, lat0, lon0, te0, re0, d0, pl0, lat1, lon1, te1, re1, d1, pl1);
}

template <int CALC>
__global__ __launch_bounds__(256) void k_rect_finalize(Frame f, const int32_t* __restrict__ hit_step, RectRec rec,
                                                       DensePlanes out) {
  stage_dm_tables();
  const size_t plane = (size_t)f.wl * f.h;
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= plane) return;
  int s = hit_step[p];
  if (s < 0) {
    store_dense_miss(out, p, plane);
    return;
  }
  const int y = (int)(p / (size_t)f.wl), x = (int)(p % (size_t)f.wl);
  store_dense(out, p, plane, rect_hit<CALC>(f, x, y, s, rec.re0[p], rec.pl0[p], rec.re1[p], rec.pl1[p]));
}

template <int CALC>
__global__ __launch_bounds__(256) void k_rect_finalize_list(Frame f, uint64_t n_hits,
                                                            const uint32_t* __restrict__ list_step,
                                                            const uint32_t* __restrict__ list_pixel, RectRec rec,
                                                            PackedHits packed) {
  stage_dm_tables();
  uint64_t k = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n_hits) return;
  if (f.n_objects && packed.color_tag[k] != ATMRT_COLOR_TERRAIN) return; // object points are already complete
  uint32_t p = list_pixel[k];
  int x = (int)(p % (uint32_t)f.wl), y = (int)(p / (uint32_t)f.wl);
  store_packed(packed, k, rect_hit<CALC>(f, x, y, (int)list_step[k], rec.re0[k], rec.pl0[k], rec.re1[k], rec.pl1[k]),
               f.p.terrain_alpha);
}


// proximity filter of one sample (TerrainData::from_lat_lon, utils.rs:74-80) over the ray's candidates: bit q = cand[q] is close.
// Any number of close objects fits (the candidate list has at most CAND_CAP = 24 entries), and the union over the two samples
// of a step is an OR; set bits ascending = object indices ascending.
static __device__ ATMRT_OBJ_FN unsigned close_mask_impl(const ObjectDev* objects, Earth e, double lat, double lon, const int* cand, int ncand) {
  const LatLonTrig t = latlon_trig(e, lat, lon);
  unsigned m = 0;
  for (int q = 0; q < ncand; q++)
    if (object_is_close(e, objects[cand[q]], t)) m |= 1u << q;
  return m;
}
static __device__ __forceinline__ unsigned close_mask(const Frame& f, const Earth& e, double lat, double lon, const int* cand,
                                                      int ncand) {
  return close_mask_impl(f.objects, e, lat, lon, cand, ncand);
}

// Rectilinear, general.  Per sample: geodesic point, terrain gather, proximity filter (TerrainData::from_lat_lon,
// utils.rs:72-88), then the step logic above.
// 4 waves per SIMD (128 VGPRs) with the object code out of line (see ATMRT_OBJ_FN above) — built with the register allocator's region
// splitting off (csrc/Makefile TRACE_RA).  With the default allocator this kernel at 128 VGPRs is wrong in object scenes with
// interprocedural register allocation already OFF (azimuth 0 for the rays without candidate objects, garbage step counts: round 4,
// full-size config 5 through 8 tiles, seed 500011 of the sweep, tools/trace_waves_probe.py): the compiler stores `direction`, `p`
// and `ncand` to their spill slots at the head of the block that follows the x_wake loop over the ray's candidates below, AHEAD of the
// exec restore, i.e. only for the lanes that have candidates.  profiles/r04/ipra/README.md part 3; `make check-isa` finds such blocks in any build.
#ifndef ATMRT_TRACE_WAVES
#define ATMRT_TRACE_WAVES 4
#endif
template <bool FILL, int CALC, bool CUBIC>
__global__ __launch_bounds__(256, ATMRT_TRACE_WAVES) void k_rect_trace(Frame f, DensePlanes out, const uint64_t* __restrict__ hit_offset,
                                                    PackedHits packed, RectRec rec, uint32_t* __restrict__ list_step,
                                                    uint32_t* __restrict__ list_pixel,
                                                    unsigned long long* __restrict__ counters,
                                                    const uint32_t* __restrict__ pixel_list, uint32_t n_list,
                                                    double* __restrict__ step_prop, OverflowArena ovf, PackedHits ovf_packed) {
  // FILL = false: count the trace points of every pixel and keep those of pixels with <= RECT_SLOTS of them in the slot arena
  // (packed / rec / list_step then are that arena, entry p * RECT_SLOTS + j).  FILL = true: write every point at its place in
  // the pixel-ordered list, for all pixels or for the listed ones (those that did not fit their slots).
  stage_dm_tables();
  const size_t plane = (size_t)f.wl * f.h;
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = pixel_list ? tid < n_list : tid < plane;
  const size_t p = pixel_list ? (live ? pixel_list[tid] : 0) : tid;
  unsigned long long steps = 0;
  if (live) {
    const Earth e = earth_for<CALC>(f);
    const int y = (int)(p / (size_t)f.wl), x = (int)(p % (size_t)f.wl);
    const bool sph = e.spherical != 0;
    const double radius = e.shape_radius;
    const bool straight = f.p.straight_rays != 0;
    const double step = f.p.simulation_step, max_dist = f.p.frame.max_distance;
    const bool terrain_opaque = f.p.terrain_alpha == 1.0;
    const double skip_above = f.tv.skip_above;
    const double alt = *f.alt;
    double direction, elevation;
    rect_ray_params(f.p, f.ph, f.c0 + x, y, direction, elevation);
    DirCalc c;
    dircalc_new(e, f.p.position.latitude, f.p.position.longitude, dm_to_degrees(direction), c);
    Stepper s;
    stepper_init(s, sph, radius, alt, elevation);
    unsigned count = 0;
    uint64_t k = FILL ? hit_offset[p] : 0;
    if (!(0.0 > max_dist || alt < -1000.0)) {
      double lat0, lon0;
      coords_at_dist(e, c, 0.0, lat0, lon0);
      double te0 = terrain_elev_or_zero(f.tv, lat0, lon0);
      int cand[CAND_CAP];
      double clo[CAND_CAP], chi[CAND_CAP];
      int ncand = 0;
      // rays with a candidate list keep the close objects of a sample as a bit mask over it; the others (more than CAND_CAP
      // candidates, or a DirectionalCalc without the pre-filter) test every object against both samples of a step.
      // x_wake: the first stepper distance at which any candidate can be close — before it the proximity filter is skipped.
      const bool use_cand = ray_candidates<CALC, CAND_CAP>(f, e, c, cand, ncand, clo, chi);
      if (!FILL && !use_cand && ncand == CAND_CAP) atomicAdd(&counters[4], 1ull); // statistics only (atmrt_last_stats)
      unsigned m0 = use_cand ? close_mask(f, e, lat0, lon0, cand, ncand) : 0u, m1 = 0u;
      double x_wake = dm_inf();
      for (int q = 0; q < ncand; q++)
        if (chi[q] >= 0.0) x_wake = clo[q] < x_wake ? clo[q] : x_wake;
      double re0 = alt, d0 = 0.0, pl0 = 0.0; // TracingState::new(.., first_path.elev, 0.0, 0.0), utils.rs:208
      double sx = 0.0, sh_ = alt, path_length = 0.0;
      bool have0 = true; // lat0 / lon0 / te0 hold the previous sample's geodesic point and terrain elevation
      for (int i = 1;; i++) {
        bool tame;
        RayState st = stepper_next<CUBIC>(s, *f.atm, sph, radius, straight, step, tame);
        if (straight) tame = __all(calc_dist_in_band(*f.atm, sh_) && calc_dist_in_band(*f.atm, st.h));
        path_length += calc_dist(sph, radius, sx, sh_, st.x, st.h, tame, f.inv_shape_radius);
        sx = st.x;
        sh_ = st.h;
        if (sx > max_dist || sh_ < -1000.0 || !(sx <= max_dist)) break; // rectilinear.rs:178 (+ NaN guard)
        // The geodesic point and the terrain lookup of a sample are skipped while nothing can need them: the ray is above every
        // post of the mosaic (ray - terrain is positive for certain, only its sign enters the crossing test) and no candidate
        // object's distance interval has been reached (no proximity test at this sample).  If the step turns out to produce
        // trace points after all, the skipped samples are evaluated then.  Rays without a candidate list test every object at
        // every sample and never skip.
        const bool awake = use_cand && sx >= x_wake; // inside (or past the start of) some candidate's interval
        bool have1 = !use_cand || awake || !(sh_ > skip_above);
        double lat1 = 0.0, lon1 = 0.0, te1 = 0.0;
        if (have1) {
          coords_at_dist(e, c, sx, lat1, lon1);
          te1 = terrain_elev_or_zero(f.tv, lat1, lon1);
        }
        m1 = 0u;
        if (awake) {
          m1 = close_mask(f, e, lat1, lon1, cand, ncand);
          x_wake = dm_inf(); // next distance of interest: the earliest start among the intervals not yet left behind
          for (int q = 0; q < ncand; q++)
            if (chi[q] >= sx) x_wake = clo[q] < x_wake ? clo[q] : x_wake;
        }
        steps++;
        StepHits hits;
        hits.n = 0;
        hits.finish = false;
        double diff1 = have0 ? re0 - te0 : 1.0, diff2 = have1 ? sh_ - te1 : 1.0;
        const bool crossing = diff1 * diff2 < 0.0;
        // the objects this step tests, ascending: union of the close lists of its two samples (utils.rs:241-250)
        unsigned m = 0u;
        bool any_object = false;
        if (use_cand) {
          m = m0 | m1;
          for (unsigned mm = m; mm; mm &= mm - 1) // drop the objects whose height band the segment does not enter
            if (object_out_of_band(f.objects[cand[__builtin_ctz(mm)]], re0, sh_)) m &= ~(mm & (0u - mm));
          any_object = m != 0u;
        } else {
          any_object = f.n_objects != 0;
        }
        if ((crossing || any_object) && !(have0 && have1)) { // the step produces trace points: its samples in full after all
#pragma unroll 1
          for (int q = 0; q < 2; q++) { // rolled: one more instance of the geodesic code, not two (rare path)
            if (q == 0 ? have0 : have1) continue;
            double la, lo;
            coords_at_dist(e, c, q == 0 ? d0 : sx, la, lo);
            const double te = terrain_elev_or_zero(f.tv, la, lo);
            if (q == 0) lat0 = la, lon0 = lo, te0 = te;
            else lat1 = la, lon1 = lo, te1 = te;
          }
          have0 = have1 = true;
          diff1 = re0 - te0;
          diff2 = sh_ - te1;
        }
        if (crossing) {
          step_push(hits, diff1 / (diff1 - diff2), -1, nullptr);
          if (terrain_opaque) hits.finish = true;
        }
        Vec3 pos1 = v3(0.0, 0.0, 0.0), pos2 = pos1;
        LatLonTrig t0{}, t1{};
        if (any_object) {
          pos1 = as_cartesian(e, lat0, lon0, re0);
          pos2 = as_cartesian(e, lat1, lon1, sh_);
          if (!use_cand) {
            t0 = latlon_trig(e, lat0, lon0);
            t1 = latlon_trig(e, lat1, lon1);
          }
        }
        auto for_each_object = [&](auto&& visit) {
          if (use_cand) {
            for (unsigned mm = m; mm; mm &= mm - 1) visit(cand[__builtin_ctz(mm)]);
          } else {
            for (int j = 0; j < f.n_objects; j++)
              if (!object_out_of_band(f.objects[j], re0, sh_) && (object_is_close(e, f.objects[j], t0) || object_is_close(e, f.objects[j], t1)))
                visit(j);
          }
        };
        if (any_object) for_each_object([&](int j) { step_object(hits, f, j, pos1, pos2); });
        if (!FILL) {
          k = (uint64_t)p * RECT_SLOTS + count;
          if (hits.n > STEP_CANDIDATES) atomicAdd(&counters[6], 1ull); // the fill pass will need Workspace::step_prop
        }
        if (FILL && hits.n > STEP_CANDIDATES) { // big step: produce the points again, straight into the list, and sort them there
          const StepGeom g{lat0, lon0, re0, d0, pl0, lat1, lon1, sh_, sx, path_length};
          const uint64_t k0 = k;
          if (crossing) big_step_put(packed, step_prop, k++, diff1 / (diff1 - diff2), nullptr, g);
          for_each_object([&](int j) { big_step_object(packed, step_prop, k, f, j, pos1, pos2, g); });
          big_step_sort(packed, step_prop, k0, hits.n);
          for (uint64_t q = k0; q < k; q++) {
            list_step[q] = (uint32_t)(i - 1);
            list_pixel[q] = (uint32_t)p;
            rec.re0[q] = re0;
            rec.pl0[q] = pl0;
            rec.re1[q] = sh_;
            rec.pl1[q] = path_length;
          }
        } else if (hits.n && (FILL || count + (unsigned)hits.n <= (unsigned)RECT_SLOTS)) {
          uint64_t k0 = k;
          step_emit(hits, packed, list_step, list_pixel, k, (uint32_t)p, i - 1, lat0, lon0, re0, d0, pl0, lat1, lon1, sh_, sx,
                    path_length);
          for (uint64_t q = k0; q < k; q++) { // terrain points: what k_rect_finalize_list needs
            rec.re0[q] = re0;
            rec.pl0[q] = pl0;
            rec.re1[q] = sh_;
            rec.pl1[q] = path_length;
          }
        }
        else if (!FILL && hits.n && hits.n <= STEP_CANDIDATES && ovf.cap) { // beyond the slots: the step's points into the overflow arena
          const unsigned long long base = atomicAdd(&counters[13], (unsigned long long)hits.n);
          if (base + (unsigned long long)hits.n <= ovf.cap) {
            uint64_t kk = base;
            step_emit(hits, ovf_packed, ovf.step, ovf.pixel, kk, (uint32_t)p, i - 1, lat0, lon0, re0, d0, pl0, lat1, lon1, sh_, sx,
                      path_length);
            for (uint64_t q = base; q < kk; q++) {
              ovf.ordinal[q] = count + (unsigned)(q - base);
              ovf.re0[q] = re0;
              ovf.pl0[q] = pl0;
              ovf.re1[q] = sh_;
              ovf.pl1[q] = path_length;
            }
          }
        }
        count += (unsigned)hits.n;
        if (hits.finish) break;
        lat0 = lat1; lon0 = lon1; te0 = te1; re0 = sh_; d0 = sx; pl0 = path_length;
        have0 = have1;
        m0 = m1;
      }
    }
    if (!FILL) {
      out.azimuth[p] = dm_to_degrees(direction);
      out.elevation_angle[p] = dm_to_degrees(elevation);
      out.hit_count[p] = count;
      if (count > (unsigned)RECT_SLOTS) atomicAdd(&counters[3], 1ull);
    }
  }
  if (!FILL) {
    steps = wave_sum(steps);
    if ((threadIdx.x & 63) == 0 && steps) atomicAdd(&counters[0], steps);
  }
}

// ---------------------------------------------------------------------------------------------
// launchers
// ---------------------------------------------------------------------------------------------

// The sliced march of a terrain-only frame: slice 0 as an ordinary grid, one host round trip for the number of groups still
// marching (it bounds the entries the later slices can write: the grid of the second kernel), then one wavefront per entry.
// false: the frame is not sliced (march_slice_layout) and the caller launches k_rect_march.
template <int MODE, bool CUBIC>
static bool launch_rect_march_sliced(const Frame& f, Workspace& ws, const DensePlanes& out, const SliceSinks& sinks, hipStream_t stream,
                                     const Frame* frame_dev = nullptr, const ObjectStepSinks* sinks_dev = nullptr) {
  SliceLayout L;
  if (!ws.slice_state || !march_slice_layout(f, L)) return false;
  const size_t n = (size_t)f.wl * f.h;
  const int slice = MARCH_SLICE_STEPS;
  SliceState st;
  char* q = ws.slice_state;
  st.calc = (DirCalc*)q; q += L.n_pad * sizeof(DirCalc);
  st.x = (double*)q; q += L.n_pad * 8;
  st.a = (double*)q; q += L.n_pad * 8;
  st.b = (double*)q; q += L.n_pad * 8;
  st.sh = (double*)q; q += L.n_pad * 8;
  st.pl = (double*)q; q += L.n_pad * 8;
  st.diff0 = (double*)q; q += L.n_pad * 8;
  st.ang = (double*)q; q += L.n_pad * 8;
  st.step = (int32_t*)q; q += L.n_pad * 4;
  st.hint = (int32_t*)q; q += L.n_pad * 4;
  st.count = (uint32_t*)q; q += L.n_pad * 4;
  st.ctl = (unsigned long long*)q; q += 64;
  st.queue = (uint32_t*)q;
  st.cap = (uint32_t)L.cap;
  st.glist = f.n_objects ? (char*)(((uintptr_t)(st.queue + L.cap) + 255) / 256 * 256) : nullptr;
  (void)hipMemsetAsync(st.ctl, 0, 64, stream);
  ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_rect_march_first<MODE, CALC, CUBIC>), dim3(cdiv(n, 256)), dim3(256), 0, stream, f,
                                                        out, sinks, (unsigned long long*)ws.counters, st, L.n_groups, slice, frame_dev, sinks_dev));
  unsigned long long ctl_host[4] = {0, 0, 0, 0};
  if (hipMemcpyAsync(ctl_host, st.ctl, sizeof ctl_host, hipMemcpyDeviceToHost, stream) != hipSuccess ||
      hipStreamSynchronize(stream) != hipSuccess)
    return true; // the error is sticky: the caller's next HIP call reports it
  // a group writes at most one entry per slice it survives
  const size_t entries = std::min<size_t>(L.cap, (size_t)ctl_host[1] * L.slices_after);
  if (entries) {
    (void)hipMemsetAsync(st.queue + ctl_host[1], 0xff, (entries - (size_t)ctl_host[1]) * sizeof(uint32_t), stream);
    ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_rect_march_cont<MODE, CALC, CUBIC>), dim3((unsigned)entries), dim3(64), 0, stream,
                                                          f, out, sinks, (unsigned long long*)ws.counters, st, L.n_groups, slice, frame_dev, sinks_dev));
  }
  hipLaunchKernelGGL(k_slice_check, dim3(1), dim3(1), 0, stream, (const unsigned long long*)st.ctl, L.n_groups,
                     (unsigned long long*)ws.counters);
  return true;
}

template <bool CUBIC>
void launch_rect_march_t(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream, hipEvent_t ev_marched) {
  size_t n = (size_t)f.wl * f.h;
  RectRec rec = carve_rec(ws.rect_rec, n);
  if (f.opaque) {
    if (!launch_rect_march_sliced<0, CUBIC>(f, ws, out, SliceSinks{ws.hit_step, rec, nullptr, OverflowArena{}, nullptr}, stream)) {
      ATMRT_LAUNCH_MARCH(0, n, stream, f, out, ws.hit_step, (const uint64_t*)nullptr, rec, (uint32_t*)nullptr, (uint32_t*)nullptr,
                         (unsigned long long*)ws.counters, (const uint32_t*)nullptr, 0u, OverflowArena{}, (const Frame*)nullptr,
                         (const ObjectStepSinks*)nullptr);
    }
    (void)hipEventRecord(ev_marched, stream);
    ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_rect_finalize<CALC>), dim3(cdiv(n, 256)), dim3(256), 0, stream,
                                                          f, ws.hit_step, rec, out));
  } else {
    RectRec slots = carve_rec(ws.slot_rec, n * RECT_SLOTS);
    const OverflowArena ovf = carve_overflow(ws.overflow_arena, ws.overflow_cap);
    if (!launch_rect_march_sliced<1, CUBIC>(f, ws, out, SliceSinks{nullptr, slots, ws.slot_step, ovf, nullptr}, stream)) {
      ATMRT_LAUNCH_MARCH(1, n, stream, f, out, ws.hit_step, (const uint64_t*)nullptr, slots, ws.slot_step, (uint32_t*)nullptr,
                         (unsigned long long*)ws.counters, (const uint32_t*)nullptr, 0u, ovf, (const Frame*)nullptr,
                         (const ObjectStepSinks*)nullptr);
    }
    (void)hipEventRecord(ev_marched, stream);
  }
}

// The crossings the counting march kept in its slots, moved to their places in the pixel-ordered list; pixels with more
// crossings than slots are collected for a second march (counters[3] was reset by the host and hands out list positions).
static __global__ __launch_bounds__(256) void k_rect_gather_slots(Frame f, const uint32_t* __restrict__ hit_count,
                                                           const uint64_t* __restrict__ hit_offset,
                                                           const uint32_t* __restrict__ slot_step, RectRec slots,
                                                           uint32_t* __restrict__ list_step, uint32_t* __restrict__ list_pixel,
                                                           RectRec rec, uint32_t* __restrict__ overflow,
                                                           unsigned long long* __restrict__ counters, int arena) {
  const size_t plane = (size_t)f.wl * f.h;
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t n = p < plane ? hit_count[p] : 0u;
  if (arena) { // the crossings beyond the slots are in the overflow arena (k_rect_scatter_overflow): the slots of every pixel count
    n = n < (uint32_t)RECT_SLOTS ? n : (uint32_t)RECT_SLOTS;
  } else {
    wave_compact_append(n > (uint32_t)RECT_SLOTS, (uint32_t)p, overflow, &counters[3]); // the pixels the second march visits
    if (n > (uint32_t)RECT_SLOTS) return;
  }
  if (p >= plane) return;
  const uint64_t k = hit_offset[p];
  for (uint32_t j = 0; j < n; j++) {
    const size_t q = (size_t)j * plane + p;
    list_step[k + j] = slot_step[q];
    list_pixel[k + j] = (uint32_t)p;
    rec.re0[k + j] = slots.re0[q];
    rec.pl0[k + j] = slots.pl0[q];
    rec.re1[k + j] = slots.re1[q];
    rec.pl1[k + j] = slots.pl1[q];
  }
}

// the arena's records to their places in the pixel-ordered list: crossing number `ordinal` of pixel p is entry hit_offset[p] + ordinal
static __global__ __launch_bounds__(256) void k_rect_scatter_overflow(uint32_t n_records, OverflowArena ovf,
                                                                      const uint64_t* __restrict__ hit_offset,
                                                                      uint32_t* __restrict__ list_step, uint32_t* __restrict__ list_pixel,
                                                                      RectRec rec) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_records) return;
  const uint32_t p = ovf.pixel[r];
  const uint64_t k = hit_offset[p] + ovf.ordinal[r];
  list_step[k] = ovf.step[r];
  list_pixel[k] = p;
  rec.re0[k] = ovf.re0[r];
  rec.pl0[k] = ovf.pl0[r];
  rec.re1[k] = ovf.re1[r];
  rec.pl1[k] = ovf.pl1[r];
}

// terrain_alpha < 1, Rectilinear: gather the recorded crossings, march the overflow pixels again listing every crossing,
// then one thread per trace point
template <bool CUBIC>
void launch_multi_fill_t(const Frame& f, Workspace& ws, uint64_t n_hits, const DensePlanes& dense, const PackedHits& packed,
                         hipStream_t stream) {
  size_t n = (size_t)f.wl * f.h;
  RectRec rec = carve_rec(ws.rect_rec, (size_t)n_hits);
  RectRec slots = carve_rec(ws.slot_rec, n * RECT_SLOTS);
  // the crossings beyond the slots: out of the overflow arena when all of them fitted it — a second march of those pixels otherwise
  // (0.7 % of the headline's pixels: a launch of one wavefront per SIMD, 12.8 ms at any frame or tile size, and most of the fill)
  const bool arena = ws.overflow_arena && ws.overflow_cap && ws.n_overflow_records <= ws.overflow_cap;
  hipLaunchKernelGGL(k_rect_gather_slots, dim3(cdiv(n, 256)), dim3(256), 0, stream, f, (const uint32_t*)dense.hit_count, ws.hit_offset,
                     ws.slot_step, slots, ws.list_step, ws.list_pixel, rec, ws.overflow, (unsigned long long*)ws.counters, arena ? 1 : 0);
  if (arena) {
    if (ws.n_overflow_records)
      hipLaunchKernelGGL(k_rect_scatter_overflow, dim3(cdiv((size_t)ws.n_overflow_records, 256)), dim3(256), 0, stream,
                         (uint32_t)ws.n_overflow_records, carve_overflow(ws.overflow_arena, ws.overflow_cap), ws.hit_offset, ws.list_step,
                         ws.list_pixel, rec);
  } else if (ws.n_overflow) {
    ATMRT_LAUNCH_MARCH(2, ws.n_overflow, stream, f, dense, ws.hit_step, ws.hit_offset, rec, ws.list_step, ws.list_pixel,
                       (unsigned long long*)ws.counters, (const uint32_t*)ws.overflow, (uint32_t)ws.n_overflow, OverflowArena{},
                       (const Frame*)nullptr, (const ObjectStepSinks*)nullptr);
  }
  if (n_hits) {
    ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_rect_finalize_list<CALC>), dim3(cdiv(n_hits, 256)), dim3(256), 0,
                                                          stream, f, n_hits, ws.list_step, ws.list_pixel, rec, packed));
  }
  launch_dense_from_packed(f, ws, packed, dense, 0, stream);
}


// the rays k_rect_march<3> left to the tracer, collected into a list (order irrelevant: every ray is independent); counters[11] = their number
static __global__ __launch_bounds__(256) void k_collect_object_rays(size_t n, const uint32_t* __restrict__ hit_count,
                                                                    uint32_t* __restrict__ list, unsigned long long* __restrict__ counters) {
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  wave_compact_append(p < n && hit_count[p] == OBJECT_RAY, (uint32_t)p, list, &counters[11]); // the rays the general tracer visits
}

// the overflow arena of a scene with objects: records + complete points
static inline OverflowArena trace_overflow_arena(const Workspace& ws) {
  OverflowArena a = carve_overflow(ws.overflow_arena, ws.overflow_cap);
  if (a.cap) a.color_tag = ws.overflow_packed.color_tag;
  return a;
}
// Scenes with objects, counting pass, phase 1: the lean march over every pixel (terrain crossings into the tracer's slot arena,
// rays that can meet an object flagged and listed); phase 2 (launch_rect_trace_objects_t, after the host has read the list's
// length) traces the listed rays with the general tracer.
template <bool CUBIC>
void launch_rect_trace_count_t(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream) {
  size_t n = (size_t)f.wl * f.h;
  RectRec slots = carve_rec(ws.slot_rec, n * RECT_SLOTS);
  // the frame and the tracer's arenas where the out-of-line object step can address them (object_step_impl): copies in HBM
  ObjectStepSinks sinks{};
  sinks.slot_packed = ws.slot_packed;
  sinks.slots = slots;
  sinks.slot_step = ws.slot_step;
  sinks.slot_pixel = ws.slot_pixel;
  sinks.ovf = trace_overflow_arena(ws);
  sinks.ovf_packed = ws.overflow_packed;
  sinks.counters = (unsigned long long*)ws.counters;
  const Frame* frame_dev = reinterpret_cast<const Frame*>(ws.step_ctx);
  const ObjectStepSinks* sinks_dev = reinterpret_cast<const ObjectStepSinks*>(ws.step_ctx + (sizeof(Frame) + 255) / 256 * 256);
  (void)hipMemcpyAsync(ws.step_ctx, &f, sizeof f, hipMemcpyHostToDevice, stream);
  (void)hipMemcpyAsync(const_cast<ObjectStepSinks*>(sinks_dev), &sinks, sizeof sinks, hipMemcpyHostToDevice, stream);
  (void)hipStreamSynchronize(stream); // both sources are on this stack frame
  // a small launch (a column tile): the time-sliced march, its groups carrying their candidate lists; else the whole grid at once
  if (!launch_rect_march_sliced<3, CUBIC>(f, ws, out, SliceSinks{ws.hit_step, slots, ws.slot_step, trace_overflow_arena(ws), ws.slot_packed.color_tag},
                                          stream, frame_dev, sinks_dev)) {
    ATMRT_LAUNCH_MARCH(3, n, stream, f, out, ws.hit_step, (const uint64_t*)nullptr, slots, ws.slot_step, ws.slot_packed.color_tag,
                       (unsigned long long*)ws.counters, (const uint32_t*)nullptr, 0u, trace_overflow_arena(ws), frame_dev, sinks_dev);
  }
  hipLaunchKernelGGL(k_collect_object_rays, dim3(cdiv(n, 256)), dim3(256), 0, stream, n, (const uint32_t*)out.hit_count, ws.object_rays,
                     (unsigned long long*)ws.counters);
}
template <bool CUBIC>
void launch_rect_trace_objects_t(const Frame& f, Workspace& ws, const DensePlanes& out, uint64_t n_rays, hipStream_t stream) {
  if (!n_rays) return;
  size_t n = (size_t)f.wl * f.h;
  RectRec slots = carve_rec(ws.slot_rec, n * RECT_SLOTS);
  ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_rect_trace<false, CALC, CUBIC>), dim3(cdiv((size_t)n_rays, 256)), dim3(256), 0, stream, f, out,
                                                        (const uint64_t*)nullptr, ws.slot_packed, slots, ws.slot_step,
                                                        ws.slot_pixel, (unsigned long long*)ws.counters,
                                                        (const uint32_t*)ws.object_rays, (uint32_t)n_rays, (double*)nullptr,
                                                        trace_overflow_arena(ws), ws.overflow_packed));
}

// Trace points kept in the slot arena by the counting pass of k_rect_trace, moved to their places in the pixel-ordered list
// (object points are complete; terrain points carry the record k_rect_finalize_list needs); overflow pixels are listed.
static __global__ __launch_bounds__(256) void k_rect_gather_trace_slots(Frame f, const uint32_t* __restrict__ hit_count,
                                                                        const uint64_t* __restrict__ hit_offset,
                                                                        const uint32_t* __restrict__ slot_step, RectRec slots,
                                                                        PackedHits sp, uint32_t* __restrict__ list_step,
                                                                        uint32_t* __restrict__ list_pixel, RectRec rec,
                                                                        PackedHits packed, uint32_t* __restrict__ overflow,
                                                                        unsigned long long* __restrict__ counters, int arena) {
  const size_t plane = (size_t)f.wl * f.h;
  const size_t p = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  uint32_t n = p < plane ? hit_count[p] : 0u;
  if (arena) {
    // the points beyond the slots are in the overflow arena: the slots of every pixel count.  (A pixel's last slots may be stale —
    // a step whose points did not all fit went to the arena whole — k_rect_scatter_trace_overflow, which runs next, overwrites them.)
    n = n < (uint32_t)RECT_SLOTS ? n : (uint32_t)RECT_SLOTS;
  } else {
    wave_compact_append(n > (uint32_t)RECT_SLOTS, (uint32_t)p, overflow, &counters[3]); // the pixels the tracer's fill pass visits
    if (n > (uint32_t)RECT_SLOTS) return;
  }
  if (p >= plane) return;
  const uint64_t k0 = hit_offset[p];
  for (uint32_t j = 0; j < n; j++) {
    const size_t q = p * RECT_SLOTS + j;
    const uint64_t k = k0 + j;
    list_step[k] = slot_step[q];
    list_pixel[k] = (uint32_t)p;
    rec.re0[k] = slots.re0[q];
    rec.pl0[k] = slots.pl0[q];
    rec.re1[k] = slots.re1[q];
    rec.pl1[k] = slots.pl1[q];
    const uint32_t tag = sp.color_tag[q];
    packed.color_tag[k] = tag;
    if (tag != ATMRT_COLOR_TERRAIN) { // terrain points are completed by k_rect_finalize_list
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

// the arena's records to their places in the pixel-ordered list (scenes with objects)
static __global__ __launch_bounds__(256) void k_rect_scatter_trace_overflow(uint32_t n_records, OverflowArena ovf, PackedHits ap,
                                                                            const int32_t* __restrict__ handed_over,
                                                                            const uint64_t* __restrict__ hit_offset,
                                                                            uint32_t* __restrict__ list_step, uint32_t* __restrict__ list_pixel,
                                                                            RectRec rec, PackedHits packed) {
  const uint32_t r = blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= n_records) return;
  const uint32_t p = ovf.pixel[r];
  uint32_t ordinal = ovf.ordinal[r];
  if (ordinal & OVERFLOW_LEAN) { // a record of the lean march: void if the ray went to the tracer afterwards
    if (handed_over[p]) return;
    ordinal &= ~OVERFLOW_LEAN;
  }
  const uint64_t k = hit_offset[p] + ordinal;
  list_step[k] = ovf.step[r];
  list_pixel[k] = p;
  rec.re0[k] = ovf.re0[r];
  rec.pl0[k] = ovf.pl0[r];
  rec.re1[k] = ovf.re1[r];
  rec.pl1[k] = ovf.pl1[r];
  const uint32_t tag = ap.color_tag[r];
  packed.color_tag[k] = tag;
  if (tag != ATMRT_COLOR_TERRAIN) { // terrain points are completed by k_rect_finalize_list
    packed.lat[k] = ap.lat[r];
    packed.lon[k] = ap.lon[r];
    packed.distance[k] = ap.distance[r];
    packed.elevation[k] = ap.elevation[r];
    packed.path_length[k] = ap.path_length[r];
    for (int c = 0; c < 3; c++) packed.normal[3 * k + c] = ap.normal[3 * r + c];
    for (int c = 0; c < 4; c++) packed.rgba[4 * k + c] = ap.rgba[4 * r + c];
  }
}

template <bool CUBIC>
void launch_rect_trace_fill_t(const Frame& f, Workspace& ws, uint64_t n_hits, const DensePlanes& dense, const PackedHits& packed,
                              hipStream_t stream) {
  size_t n = (size_t)f.wl * f.h;
  RectRec rec = carve_rec(ws.rect_rec, (size_t)n_hits);
  RectRec slots = carve_rec(ws.slot_rec, n * RECT_SLOTS);
  // the points beyond the slots: out of the overflow arena — unless it overflowed itself or a step had more points than the
  // in-register step list (those are sorted in HBM by the fill pass): then the general tracer visits those pixels a second time
  // (config 5: 22.7 ms for 1.3 % of the pixels — one wavefront per SIMD, at the speed of its dependency chain)
  const bool arena = ws.overflow_arena && ws.overflow_cap && ws.n_overflow_records <= ws.overflow_cap && !ws.step_prop;
  hipLaunchKernelGGL(k_rect_gather_trace_slots, dim3(cdiv(n, 256)), dim3(256), 0, stream, f, (const uint32_t*)dense.hit_count,
                     ws.hit_offset, ws.slot_step, slots, ws.slot_packed, ws.list_step, ws.list_pixel, rec, packed, ws.overflow,
                     (unsigned long long*)ws.counters, arena ? 1 : 0);
  if (arena) {
    if (ws.n_overflow_records)
      hipLaunchKernelGGL(k_rect_scatter_trace_overflow, dim3(cdiv((size_t)ws.n_overflow_records, 256)), dim3(256), 0, stream,
                         (uint32_t)ws.n_overflow_records, trace_overflow_arena(ws), ws.overflow_packed, (const int32_t*)ws.hit_step,
                         ws.hit_offset, ws.list_step, ws.list_pixel, rec, packed);
  } else if (ws.n_overflow) {
    ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_rect_trace<true, CALC, CUBIC>), dim3(cdiv((size_t)ws.n_overflow, 256)), dim3(256), 0,
                                                          stream, f, dense, ws.hit_offset, packed, rec, ws.list_step, ws.list_pixel,
                                                          (unsigned long long*)ws.counters, (const uint32_t*)ws.overflow,
                                                          (uint32_t)ws.n_overflow, ws.step_prop, OverflowArena{}, PackedHits{}));
  }
  if (n_hits) {
    ATMRT_DISPATCH_CALC(f.earth.calc, hipLaunchKernelGGL((k_rect_finalize_list<CALC>), dim3(cdiv(n_hits, 256)), dim3(256), 0,
                                                          stream, f, n_hits, ws.list_step, ws.list_pixel, rec, packed));
  }
  launch_dense_from_packed(f, ws, packed, dense, 0, stream);
}

// explicit instantiation of the launchers for one value of CUBIC, in two groups so that four translation units (march / trace x
// linear / spline atmospheres) compile in parallel: the lean march (k_rect_march, all modes) and the general tracer (k_rect_trace)
#define ATMRT_INSTANTIATE_MARCH(CUBIC)                                                                                        \
  template void launch_rect_march_t<CUBIC>(const Frame&, Workspace&, const DensePlanes&, hipStream_t, hipEvent_t);            \
  template void launch_multi_fill_t<CUBIC>(const Frame&, Workspace&, uint64_t, const DensePlanes&, const PackedHits&,         \
                                           hipStream_t);                                                                      \
  template void launch_rect_trace_count_t<CUBIC>(const Frame&, Workspace&, const DensePlanes&, hipStream_t);
#define ATMRT_INSTANTIATE_TRACE(CUBIC)                                                                                        \
  template void launch_rect_trace_objects_t<CUBIC>(const Frame&, Workspace&, const DensePlanes&, uint64_t, hipStream_t);      \
  template void launch_rect_trace_fill_t<CUBIC>(const Frame&, Workspace&, uint64_t, const DensePlanes&, const PackedHits&,    \
                                                hipStream_t);

} // namespace atmrt

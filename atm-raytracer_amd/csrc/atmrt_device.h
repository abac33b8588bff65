// atmrt_device.h — device-side helpers shared by the two kernel translation units (atmrt_kernels.hip: Fast,
// InterpolatingRectilinear, renderer, scan, harnesses; atmrt_march.hip: Rectilinear march and general tracer).
#pragma once

#include "atmrt_kernels.h"

namespace atmrt {

static __device__ __forceinline__ double qnan() { return __longlong_as_double(0x7ff8000000000000LL); }

// TracePoint (generators/mod.rs:21-30) of a terrain hit
struct TracePointDev {
  double lat, lon, distance, elevation, path_length;
  Vec3 normal;
};

// One bracketing pair -> interpolated terrain TracePoint (utils.rs:222-236 with :108-125)
static __device__ TracePointDev terrain_trace_point(const Frame& f, const Earth& e, double lat0, double lon0, double te0, double re0,
                                                    double dist0, double pl0, double lat1, double lon1, double te1,
                                                    double re1, double dist1, double pl1) {
  double diff1 = re0 - te0;
  double diff2 = re1 - te1;
  double prop = diff1 / (diff1 - diff2);
  Vec3 n0 = v3(0.0, 0.0, 0.0), n1 = n0;
#pragma unroll 1
  for (int k = 0; k < 2; k++) { // one instantiation of find_normal for both bracketing samples
    Vec3 n = find_normal(e, f.tv, k == 0 ? lat0 : lat1, k == 0 ? lon0 : lon1);
    if (k == 0) n0 = n;
    else n1 = n;
  }
  TracePointDev tp;
  tp.lat = lerp_ts(lat0, lat1, prop);
  tp.lon = lerp_ts(lon0, lon1, prop);
  tp.elevation = lerp_ts(te0, te1, prop);
  tp.normal = v3(lerp_ts(n0.x, n1.x, prop), lerp_ts(n0.y, n1.y, prop), lerp_ts(n0.z, n1.z, prop));
  tp.distance = lerp_ts(dist0, dist1, prop);
  tp.path_length = lerp_ts(pl0, pl1, prop);
  return tp;
}

static __device__ __forceinline__ void store_dense(const DensePlanes& o, size_t p, size_t plane, const TracePointDev& tp) {
  o.lat[p] = tp.lat;
  o.lon[p] = tp.lon;
  o.distance[p] = tp.distance;
  o.elevation[p] = tp.elevation;
  o.path_length[p] = tp.path_length;
  o.normal[p] = tp.normal.x;
  o.normal[plane + p] = tp.normal.y;
  o.normal[2 * plane + p] = tp.normal.z;
}
static __device__ __forceinline__ void store_dense_miss(const DensePlanes& o, size_t p, size_t plane) {
  double n = qnan();
  o.lat[p] = n;
  o.lon[p] = n;
  o.distance[p] = n;
  o.elevation[p] = n;
  o.path_length[p] = n;
  o.normal[p] = n;
  o.normal[plane + p] = n;
  o.normal[2 * plane + p] = n;
}
static __device__ __forceinline__ void store_packed(const PackedHits& o, uint64_t k, const TracePointDev& tp, double alpha) {
  o.lat[k] = tp.lat;
  o.lon[k] = tp.lon;
  o.distance[k] = tp.distance;
  o.elevation[k] = tp.elevation;
  o.path_length[k] = tp.path_length;
  o.normal[3 * k] = tp.normal.x;
  o.normal[3 * k + 1] = tp.normal.y;
  o.normal[3 * k + 2] = tp.normal.z;
  o.color_tag[k] = ATMRT_COLOR_TERRAIN;
  o.rgba[4 * k] = 0.0;
  o.rgba[4 * k + 1] = 0.0;
  o.rgba[4 * k + 2] = 0.0;
  o.rgba[4 * k + 3] = alpha;
}

// wave-wide sum of a 64-bit count, result valid in lane 0
static __device__ __forceinline__ unsigned long long wave_sum(unsigned long long v) {
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// Objects that can EVER be close to a sample of this ray (exact superset of Object::is_close over the whole ray).
// Spherical model: every sample, lifted to the object's elevation, lies in the great-circle plane span(pos, dir) of the
// ray's ground track, so |P - P_obj| >= distance of P_obj to that plane.  Azimuthal-equidistant model: the ground track
// is a straight line of the flat map and z differences vanish, same argument with the line.  A millimetre of slack
// covers the rounding of the recomputed sample positions.  Other models (ellipsoid geodesics, lat/lon-linear tracks):
// no pre-filter.  Returns false when the list overflowed (the caller then tests every object, still exact).
constexpr int CAND_CAP = 24;
// With lo/hi the distance interval of every candidate is returned as well: the stepper distances x at which a sample can be
// close to it.  Spherical: the angle between the sample's and the object's directions is at least their along-track angle
// difference |x/R - phi_j| and must stay below 2 asin(reach / 2(R + elev)); flat map: the along-track coordinate differs by
// less than reach.  The bounds are widened by 0.1 % + 1 mm (and to the whole ray for reaches above 1 % of the radius or
// objects below the surface by more than the radius), so they are a superset like the list itself.
// One object against one ray: can it ever be close, and (want_interval) at which stepper distances [lo, hi]?
template <int CALC>
static __device__ __forceinline__ bool candidate_interval(const Earth& e, const DirCalc& c, Vec3 nrm, const ObjectDev& o, bool want_interval,
                                                          double& lo, double& hi) {
  Vec3 rel = CALC == 2 ? o.pos : v3(o.pos.x - c.pos.x, o.pos.y - c.pos.y, 0.0);
  double dperp = dm_fabs(dot(rel, nrm));
  double reach = dm_sqrt(o.close2) + 1.0e-3;
  if (!(dperp <= reach)) return false;
  if (want_interval) {
    double along, half = reach * 1.001 + 1.0e-3;
    if (CALC == 2) {
      along = e.calc_radius * dm_atan2(dot(o.pos, c.dir), dot(o.pos, c.pos));
      if (o.elev < 0.0) half = half * (e.cart_radius / (e.cart_radius + o.elev));
      if (!(reach < 0.01 * e.cart_radius) || !(half > 0.0) || !(e.calc_radius == e.cart_radius)) half = dm_inf();
    } else {
      along = dot(rel, c.dir);
    }
    lo = along - half;
    hi = along + half;
    if (!(lo <= hi)) lo = -dm_inf(), hi = dm_inf(); // NaN anywhere: no restriction
  }
  return true;
}
template <int CALC>
static __device__ __forceinline__ bool candidates_supported(const Earth& e) {
  return (CALC == 2 && e.cart == 1) || (CALC == 0 && e.cart == 0);
}
template <int CALC>
static __device__ __forceinline__ Vec3 track_normal(const DirCalc& c) { // unit normal of the track plane / line
  return CALC == 2 ? cross(c.pos, c.dir) : v3(-c.dir.y, c.dir.x, 0.0);
}
template <int CALC, int CAP>
static __device__ __forceinline__ bool ray_candidates(const Frame& f, const Earth& e, const DirCalc& c, int* cand, int& n,
                                                      double* lo = nullptr, double* hi = nullptr) {
  n = 0;
  if (!candidates_supported<CALC>(e)) return false;
  const Vec3 nrm = track_normal<CALC>(c);
  for (int j = 0; j < f.n_objects; j++) {
    double l = 0.0, h = 0.0;
    if (candidate_interval<CALC>(e, c, nrm, f.objects[j], lo != nullptr, l, h)) {
      if (n >= CAP) return false;
      if (lo) {
        lo[n] = l;
        hi[n] = h;
      }
      cand[n++] = j;
    }
  }
  return true;
}

// A translation unit that defines DM_TABLES_LDS (see detmath.h) calls this first in every kernel: exp/log tables -> LDS.
__device__ __forceinline__ void stage_dm_tables() {
#if defined(__HIP_DEVICE_COMPILE__) && defined(DM_TABLES_LDS)
  const double* lt = &DM_LOG_TAB[0][0];
  const double* et = &DM_EXP_TAB[0][0];
  for (int i = threadIdx.x; i < 512; i += blockDim.x) DM_TABLES_LDS[i] = lt[i];
  for (int i = threadIdx.x; i < 256; i += blockDim.x) DM_TABLES_LDS[512 + i] = et[i];
  __syncthreads();
#endif
}

static inline unsigned cdiv(size_t a, size_t b) { return (unsigned)((a + b - 1) / b); }

// Wavefront compaction (the ballot / prefix primitive BASELINE's north_star names, where this path has rays to compact: the rays and
// pixels a second pass must visit): every lane with `take` appends `value` to list[*cursor ..] — ONE atomic per wavefront instead of
// one per lane, the wavefront's entries contiguous and in lane order (v_mbcnt prefix count over the ballot).  Every lane of the
// wavefront that is still running must call it (lanes that returned earlier are simply not in the ballot).
#if defined(__HIPCC__)
__device__ __forceinline__ void wave_compact_append(bool take, uint32_t value, uint32_t* __restrict__ list, unsigned long long* __restrict__ cursor) {
  const unsigned long long m = __ballot(take);
  if (!m) return; // wave-uniform
  const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u)); // takers below this lane
  const int leader = __builtin_ctzll(m);
  unsigned long long base = 0;
  if ((int)(threadIdx.x & 63) == leader) base = atomicAdd(cursor, (unsigned long long)__builtin_popcountll(m));
  base = __shfl(base, leader, 64);
  if (take) list[base + rank] = value;
}
#endif

// The DirectionalCalc kind is a compile-time constant in the heavy kernels, so only one of the four
// calculators (AzEq / FlDs / Spherical / Ellipsoid-Vincenty) is instantiated per kernel variant.
template <int CALC>
static __device__ __forceinline__ Earth earth_for(const Frame& f) {
  Earth e = f.earth;
  e.calc = CALC;
  return e;
}
#if defined(ATMRT_DEV_ONLY_SPHERICAL) // development builds (make DEV=1): only the SphericalCalc variants are compiled (1 min instead of 4)
#define ATMRT_DISPATCH_CALC(calc, STMT) \
  { constexpr int CALC = 2; STMT; }
#else
#define ATMRT_DISPATCH_CALC(calc, STMT)                  \
  switch (calc) {                                        \
    case 0: { constexpr int CALC = 0; STMT; } break;     \
    case 1: { constexpr int CALC = 1; STMT; } break;     \
    case 2: { constexpr int CALC = 2; STMT; } break;     \
    default: { constexpr int CALC = 3; STMT; } break;    \
  }
#endif

// ---------------------------------------------------------------------------------------------
// General tracer (shared part): scenes with objects (and any terrain_alpha).  get_single_pixel in full
// (utils.rs:201-289): per step the terrain sign change plus the collisions with every object that is
// close to either sample, stable-sorted by prop; stop after the step if anything opaque was hit.
// Two passes (count -> exclusive scan -> fill) because the trace-point lists have variable length.
// ---------------------------------------------------------------------------------------------
constexpr int STEP_CANDIDATES = 12; // trace points of one step kept in registers (terrain + up to 4 per object); a step with more
                                    // is counted exactly and written through the big-step route below (HBM, fill pass)

struct StepHits {
  int n;       // trace points pushed in this step; only the first STEP_CANDIDATES of them are kept below
  bool finish;
  int kind[STEP_CANDIDATES]; // -1 terrain, else object index
  Collision col[STEP_CANDIDATES];
};

static __device__ __forceinline__ void step_push(StepHits& sh, double prop, int kind, const Collision* c) {
  if (sh.n >= STEP_CANDIDATES) { // counted, not kept: the reference's step_result has no bound (utils.rs:213-282)
    sh.n++;
    return;
  }
  int j = sh.n; // step_result.sort_by(prop) is stable: insert behind every element with prop <= new prop
  while (j > 0 && sh.col[j - 1].prop > prop) {
    sh.col[j] = sh.col[j - 1];
    sh.kind[j] = sh.kind[j - 1];
    j--;
  }
  sh.kind[j] = kind;
  sh.col[j].prop = prop;
  if (c) {
    sh.col[j].normal = c->normal;
    for (int q = 0; q < 4; q++) sh.col[j].color[q] = c->color[q];
  }
  sh.n++;
}

// collisions of one object with the segment, utils.rs:251-278
static __device__ __forceinline__ bool object_out_of_band(const ObjectDev& o, double re0, double re1) {
  return (re0 < o.vlo && re1 < o.vlo) || (re0 > o.vhi && re1 > o.vhi); // false for NaN: the geometry then decides
}
#if !defined(ATMRT_OBJ_FN) // a translation unit may ask for out-of-line object code (atmrt_march_impl.h does)
#define ATMRT_OBJ_FN __forceinline__
#endif
static __device__ ATMRT_OBJ_FN void step_object_impl(StepHits& sh, const ObjectDev* objects, const uint8_t* textures, int idx, Vec3 pos1, Vec3 pos2) {
  Collision col[4];
  int nc = object_collision(objects[idx], textures, pos1, pos2, col);
  for (int q = 0; q < nc; q++) {
    if (col[q].color[3] == 0.0) continue;
    step_push(sh, col[q].prop, idx, &col[q]);
    if (col[q].color[3] == 1.0) {
      sh.finish = true;
      break;
    }
  }
}
static __device__ __forceinline__ void step_object(StepHits& sh, const Frame& f, int idx, Vec3 pos1, Vec3 pos2) {
  step_object_impl(sh, f.objects, f.textures, idx, pos1, pos2);
}

// ---- big steps: more trace points in one step than StepHits keeps ------------------------------------------------------------
// Fill pass only (the counting pass counts them exactly and raises counters[6]; such a pixel always exceeds its slots, so it is
// traced again by the fill pass).  The step's points are produced a second time, written straight to their pixel's range of the
// output list in production order with their `prop` beside them (Workspace::step_prop), and stable-sorted there by prop —
// the same order as the reference's `step_result.sort_by(prop)` over its push order (utils.rs:279).
struct StepGeom { // the two samples of the step: what an object point is interpolated from (utils.rs:261-272)
  double lat0, lon0, re0, d0, pl0, lat1, lon1, re1, d1, pl1;
};
static __device__ __noinline__ void big_step_put(const PackedHits& packed, double* __restrict__ props, uint64_t k, double prop,
                                                 const Collision* c, const StepGeom& g) {
  props[k] = prop;
  if (!c) {
    packed.color_tag[k] = ATMRT_COLOR_TERRAIN; // completed by the *_finalize_list kernels
    return;
  }
  packed.lat[k] = lerp_ts(g.lat0, g.lat1, prop);
  packed.lon[k] = lerp_ts(g.lon0, g.lon1, prop);
  packed.distance[k] = lerp_ts(g.d0, g.d1, prop);
  packed.elevation[k] = lerp_ts(g.re0, g.re1, prop);
  packed.path_length[k] = lerp_ts(g.pl0, g.pl1, prop);
  packed.normal[3 * k] = c->normal.x;
  packed.normal[3 * k + 1] = c->normal.y;
  packed.normal[3 * k + 2] = c->normal.z;
  packed.color_tag[k] = ATMRT_COLOR_RGBA;
  for (int q = 0; q < 4; q++) packed.rgba[4 * k + q] = c->color[q];
}
static __device__ __forceinline__ void big_step_object(const PackedHits& packed, double* __restrict__ props, uint64_t& k,
                                                       const Frame& f, int idx, Vec3 pos1, Vec3 pos2, const StepGeom& g) {
  Collision col[4];
  int nc = object_collision(f.objects[idx], f.textures, pos1, pos2, col);
  for (int q = 0; q < nc; q++) {
    if (col[q].color[3] == 0.0) continue;
    big_step_put(packed, props, k++, col[q].prop, &col[q], g);
    if (col[q].color[3] == 1.0) break;
  }
}
// stable insertion sort of the n points at [k0, k0 + n) by prop (rare and short: O(n^2) moves in HBM)
static __device__ __noinline__ void big_step_sort(const PackedHits& h, double* __restrict__ props, uint64_t k0, int n) {
  for (int i = 1; i < n; i++) {
    const uint64_t ki = k0 + i;
    const double prop = props[ki];
    if (!(props[ki - 1] > prop)) continue;
    const double lat = h.lat[ki], lon = h.lon[ki], dist = h.distance[ki], elev = h.elevation[ki], pl = h.path_length[ki];
    const double n0 = h.normal[3 * ki], n1 = h.normal[3 * ki + 1], n2 = h.normal[3 * ki + 2];
    const double c0 = h.rgba[4 * ki], c1 = h.rgba[4 * ki + 1], c2 = h.rgba[4 * ki + 2], c3 = h.rgba[4 * ki + 3];
    const uint32_t tag = h.color_tag[ki];
    int j = i;
    while (j > 0 && props[k0 + j - 1] > prop) {
      const uint64_t d = k0 + j, s = d - 1;
      props[d] = props[s];
      h.lat[d] = h.lat[s];
      h.lon[d] = h.lon[s];
      h.distance[d] = h.distance[s];
      h.elevation[d] = h.elevation[s];
      h.path_length[d] = h.path_length[s];
      for (int q = 0; q < 3; q++) h.normal[3 * d + q] = h.normal[3 * s + q];
      for (int q = 0; q < 4; q++) h.rgba[4 * d + q] = h.rgba[4 * s + q];
      h.color_tag[d] = h.color_tag[s];
      j--;
    }
    const uint64_t d = k0 + j;
    props[d] = prop;
    h.lat[d] = lat;
    h.lon[d] = lon;
    h.distance[d] = dist;
    h.elevation[d] = elev;
    h.path_length[d] = pl;
    h.normal[3 * d] = n0;
    h.normal[3 * d + 1] = n1;
    h.normal[3 * d + 2] = n2;
    h.rgba[4 * d] = c0;
    h.rgba[4 * d + 1] = c1;
    h.rgba[4 * d + 2] = c2;
    h.rgba[4 * d + 3] = c3;
    h.color_tag[d] = tag;
  }
}

// emit the sorted trace points of one step (fill pass).  Terrain points are finished later by the
// *_finalize_list kernels (they need find_normal); object points are complete here (utils.rs:261-272).
static __device__ __forceinline__ void step_emit(const StepHits& sh, const PackedHits& packed, uint32_t* list_step,
                                                 uint32_t* list_pixel, uint64_t& k, uint32_t pixel, int step_index,
                                                 double lat0, double lon0, double re0, double d0, double pl0, double lat1,
                                                 double lon1, double re1, double d1, double pl1) {
  for (int j = 0; j < sh.n; j++, k++) {
    list_step[k] = (uint32_t)step_index;
    list_pixel[k] = pixel;
    if (sh.kind[j] < 0) {
      packed.color_tag[k] = ATMRT_COLOR_TERRAIN;
      continue;
    }
    double prop = sh.col[j].prop;
    packed.lat[k] = lerp_ts(lat0, lat1, prop);
    packed.lon[k] = lerp_ts(lon0, lon1, prop);
    packed.distance[k] = lerp_ts(d0, d1, prop);
    packed.elevation[k] = lerp_ts(re0, re1, prop); // object hits report the RAY elevation (utils.rs:268)
    packed.path_length[k] = lerp_ts(pl0, pl1, prop);
    packed.normal[3 * k] = sh.col[j].normal.x;
    packed.normal[3 * k + 1] = sh.col[j].normal.y;
    packed.normal[3 * k + 2] = sh.col[j].normal.z;
    packed.color_tag[k] = ATMRT_COLOR_RGBA;
    for (int q = 0; q < 4; q++) packed.rgba[4 * k + q] = sh.col[j].color[q];
  }
}


// Rectilinear record: ray elevation and path length at the two samples that bracket a crossing (planar arrays)
struct RectRec {
  double* re0;
  double* pl0;
  double* re1;
  double* pl1;
};
static inline RectRec carve_rec(double* base, size_t n) {
  RectRec r;
  r.re0 = base;
  r.pl0 = base + n;
  r.re1 = base + 2 * n;
  r.pl1 = base + 3 * n;
  return r;
}

// where the out-of-line object step of the lean march (object_step_impl, atmrt_march_impl.h) writes: the general tracer's arenas of
// the counting pass.  A copy lives in HBM beside a copy of the Frame (Workspace::step_ctx): an out-of-line device function cannot
// address a kernel's by-value arguments.
struct ObjectStepSinks {
  PackedHits slot_packed;  // [plane * RECT_SLOTS]
  RectRec slots;
  uint32_t* slot_step;
  uint32_t* slot_pixel;
  OverflowArena ovf;
  PackedHits ovf_packed;
  unsigned long long* counters;
};
static_assert(sizeof(ObjectStepSinks) <= OBJECT_STEP_SINKS_MAX_BYTES, "Workspace::step_ctx reserves this much behind the Frame");

} // namespace atmrt

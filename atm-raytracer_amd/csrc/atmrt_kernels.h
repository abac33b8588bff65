// atmrt_kernels.h — launch interface between the C-ABI host code (atmrt_api.hip) and the gfx950
// kernels (atmrt_kernels.hip).  Internal; the public surface is include/atmrt.h.
#pragma once

#include <stdlib.h>
#include <string.h>

#include "atmrt_core.h"
#include "atmrt_objects.h"

namespace atmrt {

// Everything a kernel needs about the frame; passed by value (about 1.3 KB of kernel arguments).
struct Frame {
  atmrt_params_t p;
  Earth earth;
  const AtmTable* atm;      // layer table in global memory: wave-uniform indices become scalar loads
  Pinhole ph;
  TerrainView tv;
  const double* alt;        // device scalar: observer altitude after Altitude::abs (params.rs:23-30)
  const double* xs;         // xs[k] = 0 + step + ... + step (k additions): utils.rs:191-196 and the stepper's x
  const ObjectDev* objects;
  const uint8_t* textures;
  int32_t n_objects;
  int32_t n_t;              // terrain samples per column: #{k : xs[k] < max_distance}
  int32_t n_path_cap;       // path elements per row when the ray never drops below -1000 m
  int32_t c0, wl, h;        // pixel-column shard [c0, c0 + wl), image height
  int32_t opaque;           // terrain_alpha == 1.0 and no objects: at most one trace point per pixel
  // InterpolatingRectilinear: the frame is the angular lattice of Cache::get_pixel (interpolating_rectilinear.rs:80-107):
  // column x has azimuth (di0 + x) * dir_step, row y has elevation (ei0 + y) * elev_step (radians)
  int32_t lattice;
  int32_t atm_cubic;        // the atmosphere has Spline segments: launch the kernel variants that carry the quadrature path
  int32_t di0, ei0;
  double dir_step, elev_step;
  double inv_shape_radius;  // RN(1 / earth.shape_radius) (0 on a flat earth): calc_dist divides the step length by the radius (dm_div_r)
};

// column azimuth / row elevation in degrees, as handed to gen_terrain_cache / gen_path_cache
ATMRT_HD double frame_col_dir(const Frame& f, int x) {
  return f.lattice ? dm_to_degrees((double)(f.di0 + x) * f.dir_step) : fast_ray_dir(f.p, f.c0 + x);
}
ATMRT_HD double frame_row_elev(const Frame& f, int y) {
  return f.lattice ? dm_to_degrees((double)(f.ei0 + y) * f.elev_step) : fast_ray_elev(f.p, y);
}
ATMRT_HD double frame_azimuth(const Frame& f, int x) { // a single wrap into [0, 360): fast.rs:67-72, interpolating_rectilinear.rs:93-98
  double azimuth = frame_col_dir(f, x);
  if (azimuth < 0.0) azimuth += 360.0;
  else if (azimuth >= 360.0) azimuth -= 360.0;
  return azimuth;
}

// Dense per-pixel outputs ([h][wl] row-major).  `normal` is planar [3][h][wl].
struct DensePlanes {
  double* azimuth;
  double* elevation_angle;
  uint32_t* hit_count;
  double* lat;
  double* lon;
  double* distance;
  double* elevation;
  double* path_length;
  double* normal;
};

// Packed trace points (generators/mod.rs:21-30), filled in pixel order.
struct PackedHits {
  double* lat;
  double* lon;
  double* distance;
  double* elevation;
  double* path_length;
  double* normal; // [n][3]
  uint32_t* color_tag;
  double* rgba;   // [n][4]
};

// State of one row's path integration at a segment boundary (k_fast_paths runs in segments, see launch_fast_pipeline)
struct PathSegState {
  double x, a, b, px, ph, path_length;
  int32_t hint, n, done, n_final;
};

// Device counters of a frame (Workspace::counters): [0] ray-steps, [1] total of the last scan (trace points), [2] error flags,
// [3] scan total of the close lists / pixels that overflowed their slots / cursor of the overflow list, [4] rays whose candidate
// list overflowed, [5] columns whose candidate list overflowed, [6] steps with more trace points than StepHits holds,
// [7] InterpolatingRectilinear pixels with more corner points than the in-register member list, [8] their corner points
// together (size of the member arena), [9] cursor of that arena, [10] terrain lookups performed by the Rectilinear march,
// [11] rays of a scene with objects that the lean march left to the general tracer
constexpr int N_COUNTERS = 16; // [14]: ray-steps handed to the lean march's out-of-line object step; [12]: groups the time-sliced march left unfinished (must be 0: atmrt_api.hip checks); [13]: records appended to the overflow arena

// Scratch owned by the context, sized for the current frame.
// crossings per pixel recorded by the counting march (4096x2048 headline at terrain_alpha 0.5: 99.3 % of the pixels have <= 4)
constexpr int RECT_SLOTS = 4;

// Trace points beyond a pixel's RECT_SLOTS slots (translucent terrain, scenes with objects): appended by the counting passes in any
// order, each with its pixel and its ordinal among the pixel's trace points; counters[13] = records appended (more than `cap`: the
// arena is not used and the overflowing pixels are marched / traced a second time, as before round 3).  44 B per record, + a
// PackedHits entry (100 B) in scenes with objects, where a record is a complete object point or a terrain record with its tag.
// `lean_source`: set in the records of the lean march of an object scene — those of a ray the march later hands to the general
// tracer (hit_step[p] = 1 there) are void, the tracer appends that ray's points itself.
constexpr uint32_t OVERFLOW_LEAN = 0x80000000u;
struct OverflowArena {
  uint32_t *pixel, *ordinal, *step;
  double *re0, *pl0, *re1, *pl1;
  uint32_t* color_tag; // scenes with objects: the PackedHits arena's tags (the lean march writes TERRAIN), else null
  uint32_t cap;
};
constexpr size_t OBJECT_STEP_SINKS_MAX_BYTES = 1024; // ObjectStepSinks (atmrt_device.h) fits: what Workspace::step_ctx reserves behind the Frame
static inline size_t overflow_arena_bytes(size_t cap) { return cap * (3 * sizeof(uint32_t) + 4 * sizeof(double)); }
static inline OverflowArena carve_overflow(char* base, size_t cap) {
  OverflowArena a{};
  if (!base) return a;
  a.re0 = (double*)base;
  a.pl0 = a.re0 + cap;
  a.re1 = a.pl0 + cap;
  a.pl1 = a.re1 + cap;
  a.pixel = (uint32_t*)(a.pl1 + cap);
  a.ordinal = a.pixel + cap;
  a.step = a.ordinal + cap;
  a.color_tag = nullptr;
  a.cap = (uint32_t)cap;
  return a;
}

// The time-sliced march of small Rectilinear launches (atmrt_march_impl.h, k_rect_march_first / _cont): which launches take it and
// what they need.  A launch of at most MARCH_SMALL_MAX_BLOCKS 256-thread blocks is "small" (a few resident sets: column shards of a
// frame, test frames); ATMRT_MARCH_VARIANT=plain|small|sliced forces one variant for every launch (test hook: the random sweeps run
// small frames over the kernel the full-size frames use, and the other way round; same results every way).
constexpr unsigned MARCH_SMALL_MAX_BLOCKS = 16384u;
constexpr int MARCH_SLICE_STEPS = 128;
static inline int march_variant_override() {
  static const int v = [] {
    const char* e = getenv("ATMRT_MARCH_VARIANT");
    return !e ? 0 : !strcmp(e, "plain") ? 1 : !strcmp(e, "small") ? 2 : !strcmp(e, "sliced") ? 3 : 0;
  }();
  return v;
}
constexpr int WAVE_CAND = 96; // entries of a wavefront's candidate list (scenes with objects); more: every ray of the wavefront is left to the tracer
// a group's list between two slices: lo, hi, vlo, vhi (f64) and the object index (i32) of every entry, the number of entries and
// the next wake distance
constexpr size_t SLICE_GROUP_LIST_BYTES = (size_t)WAVE_CAND * (4 * sizeof(double) + sizeof(int32_t)) + 2 * sizeof(double);
struct SliceLayout {
  uint32_t n_groups;     // groups of 64 consecutive pixels
  size_t n_pad;          // pixels rounded up to whole groups
  size_t slices_after;   // an upper bound on the slices a ray can need after the first
  size_t cap;            // FIFO entries: n_groups x slices_after
  size_t bytes;          // of Workspace::slice_state
};
// false: this frame's march is not sliced (scene objects unless forced, rays one slice long, too big a launch, or forced otherwise)
static inline bool march_slice_layout(const Frame& f, SliceLayout& L) {
  const size_t n = (size_t)f.wl * f.h;
  const int ov = march_variant_override();
  if (f.p.generator != ATMRT_GEN_RECTILINEAR || n == 0 || f.n_t + 2 <= MARCH_SLICE_STEPS) return false;
  // Scenes with objects CAN be sliced (round 4: a group's candidate list travels with its state, the object steps are done out of
  // line inside the slices; bit-identical, tests/test_gpu_march_variants.py) but are not by default: config 5's tiles at 8 GPUs
  // take 8 x 43.8 ms sliced against 8 x 42.3 ms through the small-launch variant of k_rect_march<3> (the slices that contain object
  // steps run long and put their groups out of step) — ATMRT_MARCH_VARIANT=sliced forces it.
  if (f.n_objects != 0 && ov != 3) return false;
  // Translucent terrain likewise (round 4, after the march went to 5 wavefronts per SIMD; tiles of the headline at alpha 0.5, sum of the
  // tile times): 2 tiles sliced 210 ms / plain 195 / small-launch 199, 4 tiles 211 / 203 / 208, 8 tiles 224 / 235 / 218 — rays that do
  // not stop at the terrain are of near-uniform length, which is what the slices were there to even out; the launcher picks between
  // the plain and the small-launch variant by the size of the grid (ATMRT_LAUNCH_MARCH).
  if (!f.opaque && ov != 3) return false;
  if (ov ? ov != 3 : (n + 255) / 256 > MARCH_SMALL_MAX_BLOCKS) return false;
  L.n_groups = (uint32_t)((n + 63) / 64);
  L.n_pad = (size_t)L.n_groups * 64;
  L.slices_after = ((size_t)f.n_t + 2 + MARCH_SLICE_STEPS - 1) / MARCH_SLICE_STEPS;
  L.cap = (size_t)L.n_groups * L.slices_after;
  if (L.cap > 0x7fffffffull) return false; // (a frame of > 2^31 slices is marched whole)
  L.bytes = L.n_pad * (7 * sizeof(double) + 3 * sizeof(int32_t) + sizeof(DirCalc)) + 64 + L.cap * sizeof(uint32_t);
  // scenes with objects: every group's candidate list (what a wavefront of k_rect_march<3> keeps in LDS) travels with its state
  if (f.n_objects) L.bytes += 256 + (size_t)L.n_groups * SLICE_GROUP_LIST_BYTES;
  return true;
}

struct Workspace {
  double* alt;            // [1]
  DirCalc* colcalc;       // [wl]   Fast: per-column DirectionalCalc
  double* prof;           // [n_t][wl] Fast: terrain profile, sample-major so a wavefront reads 64 columns coalesced
  double* pelev;          // [h][n_path_cap] Fast: ray elevation per row
  double* plen;           // [h][n_path_cap] Fast: running path length per row
  int32_t* npath;         // [h]
  PathSegState* path_seg; // [h] Fast: integration state between path segments
  double* dprev;          // [h][wl] Fast: ray-minus-terrain difference at the last sample of the previous intersect segment
  double* pelev_t;        // [n_path_cap][h] scenes with objects: pelev / plen sample-major for k_fast_trace (lanes = rows)
  double* plen_t;
  int32_t* hit_step;      // [h][wl] first hit: index of the older sample of the pair, or -1
  uint64_t* hit_offset;   // [h][wl] exclusive scan of hit_count
  uint64_t* scan_tmp;     // block sums for the scan
  uint64_t* counters;     // [0] ray-steps, [1] total hits
  uint32_t* list_step;    // multi-hit: per trace point, the step index and ...
  uint32_t* list_pixel;   // ... its pixel
  double* rect_rec;       // Rectilinear: [4][n] ray elevation / path length at the two bracketing samples
  // Rectilinear with translucent terrain: the counting march keeps the first RECT_SLOTS crossings of every pixel, so that only
  // pixels with more crossings are marched a second time
  uint32_t* slot_step;    // [RECT_SLOTS][h][wl]
  double* slot_rec;       // [4][RECT_SLOTS][h][wl]
  uint32_t* overflow;     // pixels with more than RECT_SLOTS crossings
  uint32_t* slot_pixel;   // scenes with objects: [h][wl][RECT_SLOTS] (written by step_emit, not read)
  PackedHits slot_packed; // scenes with objects: trace points of the slots, entry p * RECT_SLOTS + j
  uint64_t n_overflow;    // their number (host copy of counters[3] after the counting march)
  // scenes with objects (Fast): geodesic point of every sample and the objects close to it (utils.rs:74-80)
  double* plat;           // [n_t][wl]
  double* plon;           // [n_t][wl]
  uint32_t* ccount;       // [n_t][wl] number of close objects
  uint64_t* coffset;      // [n_t][wl] exclusive scan of ccount
  uint32_t* clist;        // object indices, ascending per sample
  int32_t* col_cand;      // [wl][64] objects that can be close to any sample of the column (ascending), and ...
  int32_t* col_ncand;     // ... their number; -1 = no list, test every object
  double* col_lo;         // [wl][64] distances between which a sample of the column can be close to the candidate ...
  double* col_hi;
  uint8_t* traced;        // [h][wl] Fast with objects: 1 = the pixel can have a step with an object (k_fast_flag_rows)
  uint32_t* object_rays;  // Rectilinear, scenes with objects: pixels the lean march left to the general tracer
  char* step_ctx;         // Rectilinear, scenes with objects: Frame + ObjectStepSinks in HBM for the lean march's out-of-line object step
  double* step_prop;      // fill pass, frames with big steps only: `prop` of every listed trace point (big_step_sort)
  uint32_t* px_steps;     // optional [h][wl]: ray-steps of each pixel (InterpolatingRectilinear counts referenced lattice pixels only)
  char* overflow_arena;   // Rectilinear, translucent terrain or objects: trace points beyond the slots (OverflowArena), or null
  PackedHits overflow_packed; // scenes with objects: the arena's complete points
  size_t overflow_cap;    // its capacity in records
  uint64_t n_overflow_records; // host copy of counters[13] after the counting march
  char* slice_state;      // time-sliced march (march_slice_layout): ray state between two slices + the FIFO of groups, or null
};

// InterpolatingRectilinear scratch
struct InterpBuffers {
  double* dir;        // [H][W] ray_params_table.direction (radians), full image
  double* elev;       // [H][W]
  double* colmin;     // [W]
  double* rowmin;     // [H]
  int32_t* key_e;     // [h][wl] elev_index of the pixel's first lattice corner
  int32_t* key_d;     // [h][wl]
  double* rem_e;      // [h][wl]
  double* rem_d;      // [h][wl]
  int32_t* bounds;    // [4] min/max of elev_index and dir_index over the shard
  uint8_t* referenced;// [ne][nd]
};
struct LatticeResult { // the lattice frame's packed result
  const uint32_t* hit_count;
  const uint64_t* hit_offset;
  const double* azimuth;
  const double* elevation_angle;
  PackedHits hits;
  const uint32_t* px_steps;
  int32_t nd, ne;
};
struct BlendArena { // member lists of the pixels k_interp_blend_big blends (more than 4 corner points together)
  uint64_t* k;
  double* dist;
  uint32_t* group;
  uint8_t* corner;
  uint8_t* tag;
};
void launch_interp_blend_big(const Frame& f, Workspace& ws, const InterpBuffers& ib, const LatticeResult& lr, bool fill,
                             const DensePlanes& dense, const PackedHits& packed, const BlendArena& arena, hipStream_t stream);
void launch_fov_table(const Frame& f, const InterpBuffers& ib, hipStream_t stream);
void launch_lattice_keys(const Frame& f, const InterpBuffers& ib, double min_elev_step, double min_dir_step, hipStream_t stream);
void launch_interp_blend(const Frame& f, Workspace& ws, const InterpBuffers& ib, const LatticeResult& lr, bool fill,
                         const DensePlanes& dense, const PackedHits& packed, hipStream_t stream);
void launch_interp_finish(const Frame& f, Workspace& ws, const InterpBuffers& ib, const LatticeResult& lr,
                          const DensePlanes& dense, const PackedHits& packed, hipStream_t stream);

// All launches go to `stream`; none of them synchronises or allocates.
void launch_resolve(const Frame& f, Workspace& ws, ObjectDev* objects_mut, hipStream_t stream);
// scenes with objects / translucent terrain + objects: general tracer (count -> scan -> fill)
void launch_fast_profile_ll(const Frame& f, Workspace& ws, hipStream_t stream);
void launch_close_count(const Frame& f, Workspace& ws, hipStream_t stream);
void launch_close_fill(const Frame& f, Workspace& ws, hipStream_t stream);
void launch_scan_u32(const uint32_t* in, size_t n, uint64_t* tmp, uint64_t* out, unsigned long long* total, hipStream_t stream);
void launch_trace_count(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream);
void launch_trace_fill(const Frame& f, Workspace& ws, uint64_t n_hits, const DensePlanes& dense, const PackedHits& packed,
                       hipStream_t stream);
void launch_fast_paths(const Frame& f, Workspace& ws, hipStream_t stream, int i_begin, int i_end); // atmrt_paths.hip
#ifndef ATMRT_FAST_SEGMENTS
#define ATMRT_FAST_SEGMENTS 4
#endif
constexpr int FAST_SEGMENTS = ATMRT_FAST_SEGMENTS;
int launch_fast_pipeline(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream, hipStream_t stream2,
                         hipEvent_t ev_fork, hipEvent_t* ev_seg, hipEvent_t* ev_scan, hipEvent_t* timing); // returns the number of segments
void launch_fast_caches(const Frame& f, Workspace& ws, hipStream_t stream, hipStream_t stream2, hipEvent_t ev,
                        hipEvent_t ev_join, hipEvent_t* timing /* [0..1] phase A, [2..3] phase B */);
void launch_fast_intersect(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream);
void launch_fast_finalize(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream);
void launch_rect_march(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream, hipEvent_t ev_marched);
void launch_scan_counts(const Frame& f, Workspace& ws, const uint32_t* hit_count, hipStream_t stream);
void launch_pack_first_hits(const Frame& f, Workspace& ws, const DensePlanes& dense, const PackedHits& packed,
                            hipStream_t stream);
// multi-hit (terrain_alpha < 1): count -> scan -> fill
void launch_multi_fill(const Frame& f, Workspace& ws, uint64_t n_hits, const DensePlanes& dense, const PackedHits& packed,
                       hipStream_t stream);

void launch_multi_fill_fast(const Frame& f, Workspace& ws, uint64_t n_hits, const DensePlanes& dense,
                            const PackedHits& packed, hipStream_t stream);

void launch_draw_image(size_t n_pixels, const atmrt_coloring_t& col, double terrain_alpha, bool packed_valid,
                       const uint32_t* hit_count, const uint64_t* hit_offset, const PackedHits& hits, const DensePlanes& dense,
                       uint8_t* rgb, hipStream_t stream);

void launch_rect_trace_count(const Frame& f, Workspace& ws, const DensePlanes& out, hipStream_t stream);
void launch_rect_trace_objects(const Frame& f, Workspace& ws, const DensePlanes& out, uint64_t n_rays, hipStream_t stream);
void launch_rect_trace_fill(const Frame& f, Workspace& ws, uint64_t n_hits, const DensePlanes& dense, const PackedHits& packed,
                            hipStream_t stream);
void launch_dense_from_packed(const Frame& f, Workspace& ws, const PackedHits& packed, const DensePlanes& dense, int fast_angles,
                              hipStream_t stream);

// harness kernels (diagnostic subcommands of the reference)
void launch_get_elev(const Frame& f, size_t n, const double* lat, const double* lon, double* elev, uint8_t* valid,
                     hipStream_t stream);
void launch_ray_paths(const Frame& f, double h0, size_t n_angles, const double* angles_deg, int straight, double step,
                      size_t n_steps, double* x, double* h, hipStream_t stream);
void launch_atm_sample(const Frame& f, size_t n, const double* alt, double* t, double* p, double* nidx, double* dn,
                       hipStream_t stream);
void launch_math_probe(int op, size_t n, const double* a, const double* b, double* out0, double* out1, hipStream_t stream);
void launch_coords_at_dist(const Frame& f, double lat0, double lon0, double dir, size_t n, const double* dist,
                           double* lat, double* lon, hipStream_t stream);

} // namespace atmrt

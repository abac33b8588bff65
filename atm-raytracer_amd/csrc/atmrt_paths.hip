// atmrt_paths.hip — phase B of the Fast generator: the per-row ray paths (gen_path_cache, utils.rs:136-174) on gfx950.
// A translation unit of its own so that detmath's exp/log tables can live in LDS here (DM_TABLES_LDS): the kernel is one
// long dependent chain per ray, and every table look-up of an n(h) evaluation sits on that chain.
#if defined(__HIP_DEVICE_COMPILE__)
#define DM_TABLES_LDS atmrt_dm_tables_lds
__shared__ double atmrt_dm_tables_lds[768];
#endif
#include "atmrt_kernels.h"
#include "atmrt_device.h"

namespace atmrt {

// Phase B.  Rows are independent and each ray is a sequential RK4 chain (H-way parallelism only), so the kernel is
// latency-bound.  Two facts shorten the chain without touching the arithmetic:
//   * the three n(h) evaluations behind one right-hand side (n(h), n(h - eps), n(h + eps) of Environment::n / dn) are
//     independent of each other;
//   * n and dn depend only on the POSITION argument of an RK4 stage, and stage 2's position a + d/2 k1a (k1a = b) is known
//     when the step starts, likewise stage 4's position once k1b is known — so stages 1 and 2, then stages 3 and 4, can
//     evaluate their refractive indices concurrently.
// Each ray therefore gets an OCTET of lanes: lanes 0-2 serve stages 1 and 3, lanes 4-6 stages 2 and 4 (lanes 3 and 7
// duplicate); values are exchanged with shuffles and the cheap remainder is computed redundantly by all eight lanes.
// Same operations in the same order per value; a quarter of the dependent chain of the one-lane-per-ray version.
// Data movement inside an octet without LDS: a DPP `quad_perm` broadcast hands lane k of every quad to the quad's four lanes, and
// `row_half_mirror` (lane i <-> lane 7 - i of each half row = octet) hands a value that is uniform within each quad to the other
// quad.  VALU-speed moves on the dependent chain where __shfl (ds_bpermute: an LDS round trip) used to be.
template <int CTRL>
__device__ __forceinline__ double dpp_move(double v) {
  const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), CTRL, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), CTRL, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
constexpr int DPP_QUAD_BCAST0 = 0x00, DPP_QUAD_BCAST1 = 0x55, DPP_QUAD_BCAST2 = 0xAA, DPP_ROW_HALF_MIRROR = 0x141;

template <bool CUBIC>
struct OctetRK4 {
  const AtmTable& atm;
  int sub, base; // lane within the octet, first lane of the octet within the wavefront

  // n and dn at position `pos` for both stage groups: this lane evaluates its share, everyone receives both results
  __device__ __forceinline__ void eval(bool spherical, double radius, double pos_a, double pos_b, int& hint, AtmLayerCache& cache,
                                       double& n_a, double& dn_a, double& n_b, double& dn_b, bool& certified) const {
    const double eps = 0.01;
    const double pos = (sub & 4) ? pos_b : pos_a;
    const double h = spherical ? pos - radius : pos;
    const int e = sub & 3;
    const double hh = e == 1 ? h - eps : e == 2 ? h + eps : h;
    const double nv = refr_n_speculative<CUBIC>(atm, cache, hh, hint, certified);
    // every lane of a quad: its quad's n(h), n(h - eps), n(h + eps); then dn for the quad's stage, then the other quad's pair
    const double n_q = dpp_move<DPP_QUAD_BCAST0>(nv);
    const double q1 = dpp_move<DPP_QUAD_BCAST1>(nv), q2 = dpp_move<DPP_QUAD_BCAST2>(nv);
    // shortcut divisions whether or not every lane was certified: next() discards the step if one was not
    const double dn_q = (q2 - q1) * REFR_INV_2EPS;
    const double n_o = dpp_move<DPP_ROW_HALF_MIRROR>(n_q), dn_o = dpp_move<DPP_ROW_HALF_MIRROR>(dn_q);
    const bool second = (sub & 4) != 0; // this lane's quad serves stage group b
    n_a = second ? n_o : n_q;
    dn_a = second ? dn_o : dn_q;
    n_b = second ? n_q : n_o;
    dn_b = second ? dn_q : dn_o;
  }
  static __device__ __forceinline__ double accel(bool spherical, double a, double b, double n, double dn) {
    return accel_rhs<true>(spherical, a, b, n, dn);
  }

  // PathStepper::next, identical in value to stepper_next_with; tame: as stepper_next_with reports it.
  // The step is computed with the shortcut divisions throughout and their contract is verified after the fact: every n of every
  // lane from a certified interval (eval's flag) and the four stage slopes at most 2^100 — stage i is exact if its slope k_ia is in range,
  // k_1a is the start state and each later slope comes from exact earlier stages, so if all four pass all four stages were exact.
  // Otherwise (a wavefront straddling a layer boundary for a few steps; a pathological atmosphere) the step is thrown away and
  // repeated by the serial stepper with its per-stage guards.  One vote per step on this kernel's dependent chain.
  __device__ __forceinline__ RayState next(Stepper& s, AtmLayerCache& cache, bool spherical, double radius, bool straight, double step,
                                           bool& tame) const {
    if (straight) return stepper_next_with(s, spherical, radius, true, step, SerialAccel<CUBIC>{atm}, tame);
    const double d = spherical ? step / radius : step;
    const double half = 0.5 * d, sixth = d / 6.0;
    const double a = s.a, b = s.b;
    const int hint0 = s.hint;
    double n1, dn1, n2, dn2, n3, dn3, n4, dn4;
    bool cert12, cert34; // per lane
    const double k1a = b;
    const double a2 = DM_FMA(half, k1a, a);
    eval(spherical, radius, a, a2, s.hint, cache, n1, dn1, n2, dn2, cert12);
    const double k1b = accel(spherical, a, b, n1, dn1);
    const double k2a = DM_FMA(half, k1b, b);
    const double k2b = accel(spherical, a2, k2a, n2, dn2);
    const double k3a = DM_FMA(half, k2b, b);
    const double a3 = DM_FMA(half, k2a, a), a4 = DM_FMA(d, k3a, a);
    eval(spherical, radius, a3, a4, s.hint, cache, n3, dn3, n4, dn4, cert34);
    const double k3b = accel(spherical, a3, k3a, n3, dn3);
    const double k4a = DM_FMA(d, k3b, b);
    const double k4b = accel(spherical, a4, k4a, n4, dn4);
    const bool slopes_ok = !(dm_fabs(k1a) > ACCEL_FAST_MAX_B) && !(dm_fabs(k2a) > ACCEL_FAST_MAX_B) && !(dm_fabs(k3a) > ACCEL_FAST_MAX_B) &&
                           !(dm_fabs(k4a) > ACCEL_FAST_MAX_B);
    if (!__all(slopes_ok && cert12 && cert34)) {
      s.hint = hint0;
      return stepper_next_with(s, spherical, radius, false, step, SerialAccel<CUBIC>{atm}, tame);
    }
    tame = true;
    s.a = rk4_sum(a, sixth, k1a, k2a, k3a, k4a);
    s.b = rk4_sum(b, sixth, k1b, k2b, k3b, k4b);
    s.x = s.x + step;
    RayState out;
    out.x = s.x;
    if (spherical) {
      out.h = s.a - radius;
      out.dh = s.b / radius;
    } else {
      out.h = s.a;
      out.dh = s.b;
    }
    return out;
  }
};

constexpr int PATH_LANES = 8; // lanes per ray

template <bool CUBIC>
__global__ __launch_bounds__(64) void k_fast_paths(Frame f, double* __restrict__ pelev, double* __restrict__ plen,
                                                   int32_t* __restrict__ npath, PathSegState* __restrict__ seg, int i_begin,
                                                   int i_end) {
  // Steps i_begin .. i_end - 1 of every row.  The frame's paths are integrated in a few segments so that the intersect scan of
  // the samples already written can run (on the other stream) while the next segment is integrated; a row's state at a
  // segment boundary travels through `seg`.
  stage_dm_tables();
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int sub = t & (PATH_LANES - 1);
  const int row = t / PATH_LANES;
  const int y = row < f.h ? row : f.h - 1; // surplus octets repeat the last row (every lane must reach the shuffles)
  const bool writer = sub == 0 && row < f.h;
  const bool sph = f.earth.spherical != 0;
  const double radius = f.earth.shape_radius;
  const bool straight = f.p.straight_rays != 0;
  const double step = f.p.simulation_step, max_dist = f.p.frame.max_distance;
  const double alt = *f.alt;
  const OctetRK4<CUBIC> rk4{*f.atm, sub, (int)(threadIdx.x & 63 & ~(PATH_LANES - 1))};
  Stepper s;
  stepper_init(s, sph, radius, alt, dm_to_radians(frame_row_elev(f, y)));
  size_t base = (size_t)y * f.n_path_cap;
  if (writer && i_begin <= 1) {
    pelev[base] = alt;
    plen[base] = 0.0;
  }
  int n = 1;
  double px = 0.0, ph = alt, path_length = 0.0;
  bool done = false;
  int n_final = 0;
  if (i_begin > 1) { // resume: every lane of the octet (and a surplus octet repeating the last row) reads the row's state
    const PathSegState st0 = seg[y];
    s.x = st0.x;
    s.a = st0.a;
    s.b = st0.b;
    s.hint = st0.hint;
    px = st0.px;
    ph = st0.ph;
    path_length = st0.path_length;
    n = st0.n;
    done = st0.done != 0;
    n_final = st0.n_final;
  }
  // utils.rs:159-171: push, then stop once the PREVIOUS state is beyond max_distance or below -1000 m.
  // The loop bound is wave-uniform; an octet that has finished keeps stepping without storing, so that every lane of
  // the wavefront takes part in every shuffle.
  const int i_last = i_end < f.n_path_cap ? i_end : f.n_path_cap;
  AtmLayerCache cache; // the hinted layer's parameters, wave-uniform, read again only when the layer changes
  for (int i = i_begin; i < i_last; i++) {
    bool tame;
    RayState st = rk4.next(s, cache, sph, radius, straight, step, tame);
    if (straight) tame = __all(calc_dist_in_band(*f.atm, ph) && calc_dist_in_band(*f.atm, st.h));
    path_length += calc_dist(sph, radius, px, ph, st.x, st.h, tame); // (dm_div here: dm_div_r's corrections measured 0.08 ms slower on this kernel's chain)
    if (!done) {
      if (writer) {
        pelev[base + n] = st.h;
        plen[base + n] = path_length;
      }
      n++;
      if (px > max_dist || ph < -1000.0) {
        done = true;
        n_final = n;
      }
    }
    px = st.x;
    ph = st.h;
    if (__all(done)) break;
  }
  if (writer) {
    npath[y] = done ? n_final : n; // elements written so far; final after the last segment (utils.rs:160-170)
    PathSegState st1;
    st1.x = s.x;
    st1.a = s.a;
    st1.b = s.b;
    st1.px = px;
    st1.ph = ph;
    st1.path_length = path_length;
    st1.hint = s.hint;
    st1.n = n;
    st1.done = done ? 1 : 0;
    st1.n_final = n_final;
    seg[y] = st1;
  }
}

void launch_fast_paths(const Frame& f, Workspace& ws, hipStream_t stream, int i_begin, int i_end) {
  if (f.atm_cubic)
    hipLaunchKernelGGL((k_fast_paths<true>), dim3(cdiv((size_t)f.h * PATH_LANES, 64)), dim3(64), 0, stream, f, ws.pelev, ws.plen, ws.npath,
                       ws.path_seg, i_begin, i_end);
  else
    hipLaunchKernelGGL((k_fast_paths<false>), dim3(cdiv((size_t)f.h * PATH_LANES, 64)), dim3(64), 0, stream, f, ws.pelev, ws.plen, ws.npath,
                       ws.path_seg, i_begin, i_end);
}

} // namespace atmrt

// atmrt_multi.hip — the multi-GPU path BELOW the C ABI (include/atmrt.h "several GPUs of one node").
//
// The reference renders a frame with one call, `generator.generate()` (src/generator/mod.rs:72-86; trait Generator,
// generators/mod.rs:82-84), and fills `result[y][x]` row by row (fast.rs:52-92).  Pixels are independent (rectilinear.rs:32-37), so
// here the image is cut into pixel-column tiles, one per device / rank, every device marches its tile against its own copy of the
// terrain mosaic, and the ONLY exchange is the finished frame:
//
//   host consumer   (atmrt_generate on a multi-device context): every device copies its tile's planes straight into the one
//                   page-locked [H][W] block with strided device-to-host copies over its own PCIe link; the variable-length
//                   trace-point lists go through per-device staging and are merged row segment by row segment on the devices'
//                   host threads.  No collective: nothing needs the frame on a GPU.
//   device consumer (atmrt_generate_image_device): the tile's nine planes live in ONE slab (84 B/pixel), one ncclAllGather
//                   (RCCL over xGMI, hand-written against <rccl/rccl.h>) moves every slab to every device, and k_assemble_image
//                   permutes the rank-major slabs into the row-major [H][W] planes.  Lists: count -> scan -> offset on the device
//                   (atmrt_image_hits_device: one 8-byte all-gather of the totals, one all-gather of the packed lists, G + 1 scans
//                   and one gather kernel).
//
// xGMI is point-to-point (7 links of ~153 GB/s per GPU): with 8 ranks a tile is one link's worth per peer, so the direct
// all-gather is bound by a single link — 92 MB per tile at 4096x2048 is ~0.6 ms at link speed against ~35 ms of marching.
//
// One code path serves three launch models: one process per GPU (atmrt_ctx_comm_init_rank), one process with a thread per device
// (atmrt_ctx_create_multi), and a host-supplied transport (atmrt_ctx_comm_init_external).  RCCL is loaded with dlopen on first
// use, so a single-GPU host needs no RCCL and a process that already carries one (PyTorch's) shares it.
#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <thread>

#include "atmrt_hostmem.h"
#include "atmrt_multi.h"

using namespace atmrt;

// ---------------------------------------------------------------------------------------------
// RCCL, resolved at run time
// ---------------------------------------------------------------------------------------------
namespace {

struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommInitAll) CommInitAll = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommAbort) CommAbort = nullptr; // optional: only the watchdog of the probe collective uses it
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  std::string error;
};

Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* env = getenv("ATMRT_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n || !*n) continue;
      r.handle = dlopen(n, RTLD_NOW | RTLD_LOCAL);
      if (r.handle) break;
      r.error = dlerror();
    }
    if (!r.handle) return;
    auto sym = [&](const char* name) {
      void* p = dlsym(r.handle, name);
      if (!p) r.error = std::string("librccl lacks ") + name;
      return p;
    };
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(sym("ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(sym("ncclCommInitRank"));
    r.CommInitAll = reinterpret_cast<decltype(r.CommInitAll)>(sym("ncclCommInitAll"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(sym("ncclAllGather"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(sym("ncclCommDestroy"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(sym("ncclGetErrorString"));
    r.CommAbort = reinterpret_cast<decltype(r.CommAbort)>(dlsym(r.handle, "ncclCommAbort"));
    if (!r.GetUniqueId || !r.CommInitRank || !r.CommInitAll || !r.AllGather || !r.CommDestroy || !r.GetErrorString) {
      dlclose(r.handle);
      r.handle = nullptr;
    } else {
      r.error.clear();
    }
  });
  return &r;
}

// grow-only page-locked host buffer (staging of the host-consumer route and of the external transport)
struct HostBuf {
  void* ptr = nullptr;
  size_t cap = 0;
  hipError_t reserve(size_t bytes) {
    if (bytes <= cap) return hipSuccess;
    release();
    const size_t want = bytes + bytes / 8 + 256;
    hipError_t e = hipHostMalloc(&ptr, want, hipHostMallocPortable);
    if (e == hipSuccess) cap = want;
    else ptr = nullptr;
    return e;
  }
  void release() {
    if (ptr) (void)hipHostFree(ptr);
    ptr = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const { return static_cast<T*>(ptr); }
  ~HostBuf() { release(); }
};

constexpr int N_F64_PLANES = 10; // azimuth, elevation_angle, lat, lon, distance, elevation, path_length, normal x / y / z

size_t pad256(size_t b) { return (b + 255) / 256 * 256; }

} // namespace

// ---------------------------------------------------------------------------------------------
// a rank's place in the frame
// ---------------------------------------------------------------------------------------------
namespace atmrt {

// What every rank appends to its slab: the one collective of a frame also tells every rank how many trace points each tile holds
// (so the lists need no exchange of totals — and no rank can disagree about whether a second collective happens), which frame the
// slab belongs to (ranks out of step are an error, not a silently mixed image) and how long the tile took (the next frame's tiling).
struct SlabTrailer {
  uint64_t n_hits;  // trace points of the tile's packed lists (0 for a first-hit frame)
  uint64_t seq;     // frames this rank has exchanged before this one
  double tile_ms;   // device time of the tile's generate
  uint32_t packed;  // 1: the frame has lists
  uint32_t c0, c1;  // the tile's columns as this rank sees them
  uint32_t _pad;
};
static_assert(sizeof(SlabTrailer) == 40, "the trailer travels as bytes");
constexpr size_t SLAB_TRAILER_BYTES = 256; // its own 256-byte line at the end of the slab

struct Comm {
  int rank = 0, world = 1;
  int route = ATMRT_ROUTE_NONE;
  ncclComm_t nccl = nullptr;
  MultiGroup* group = nullptr; // the devices of one process (peer copies, host threads)
  atmrt_all_gather_fn ext = nullptr;
  atmrt_all_gather_device_fn ext_dev = nullptr;
  void* ext_user = nullptr;
  hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr, ev_a1 = nullptr;
  DevBuf d_slab, d_gathered, d_small, d_hits_send, d_hits_recv, d_loc_off, d_scan_tmp, d_rgb_tile, d_rgb_all, d_tiling;
  HostBuf h_send, h_recv, h_stage;
  // The pixel-column tiles.  `cols` (world + 1 boundaries) is the tiling the NEXT frame will use: equal widths to begin with
  // (shard_begin), then — tile_rebalance — widths that equalise the tiles' times, computed by every rank from the same gathered
  // trailers and therefore identical on all of them.  `cols_frame` is the tiling of the last frame that was exchanged (what the
  // assembly kernels and the lists read; a copy lives in d_tiling).
  std::vector<int> cols, cols_frame;
  int cols_W = -1;
  bool balance = false;      // re-tile after a frame whose slowest tile took more than 1 % longer than the mean
  bool cols_pinned = false;  // atmrt_debug_set_tiling: the caller's tiling stays
  // geometry of the last frame
  int W = 0, H = 0, wl_max = 0;
  size_t slab_bytes = 0;
  DensePlanes slab_planes{};
  DensePlanes last_image{}; // where the last frame's [H][W] planes were assembled (caller-owned)
  bool image_valid = false;
  bool exchanged = false;   // the last frame went through its collective on this rank (and so, by construction, on every rank)
  uint64_t seq = 0;         // frames exchanged so far
  std::vector<SlabTrailer> trailers; // of the last exchanged frame, one per rank
  int fail_countdown = 0;   // atmrt_debug_fail_next_collective: the n-th collective from now fails on this rank
  atmrt_comm_timings_t tm{};
  // host-consumer route: this device's row totals and staging layout
  std::vector<uint64_t> row_total;
};

struct MultiGroup {
  std::vector<atmrt_ctx*> kids;
  std::vector<int> devices;
  std::vector<std::thread> workers;
  std::mutex m;
  std::condition_variable cv_task, cv_done;
  uint64_t epoch = 0;
  int pending = 0;
  bool stop = false;
  const std::function<int(atmrt_ctx*, int)>* task = nullptr;
  std::vector<int> rc;
  // barrier of the peer-copy route (every worker is inside the same task when it is used)
  std::mutex bm;
  std::condition_variable bcv;
  int bcount = 0;
  uint64_t bgen = 0;
  std::vector<void*> recv_ptr;
  std::vector<ncclComm_t> nccl;
  atmrt_comm_timings_t tm{};

  std::string demoted;  // why the context left the RCCL route for peer copies (empty: it did not)
  bool aborted = false; // a device failed inside a task that has barriers: the others stop waiting (guarded by bm)

  // All devices meet here; false: one of them failed before they all had arrived, and nobody waits any more.  A barrier that
  // COMPLETED stays completed: a device that fails right after leaving it (its own arguments were wrong: tile_hits reports that
  // after the collective) sets `aborted` while its peers may not yet have woken up from the same barrier — they must go on.
  bool barrier() {
    std::unique_lock<std::mutex> lk(bm);
    if (aborted) return false;
    const uint64_t gen = bgen;
    if (++bcount == (int)kids.size()) {
      bcount = 0;
      bgen++;
      bcv.notify_all();
      return true;
    }
    bcv.wait(lk, [&] { return bgen != gen || aborted; });
    return bgen != gen;
  }
  void abort_task() {
    std::lock_guard<std::mutex> lk(bm);
    aborted = true;
    bcv.notify_all();
  }

  // Runs fn(child, index) on every device's own host thread and waits; returns the first failure.
  int run(const std::function<int(atmrt_ctx*, int)>& fn) {
    {
      std::lock_guard<std::mutex> bl(bm); // every worker is idle here: a fresh barrier for the new task
      aborted = false;
      bcount = 0;
    }
    std::unique_lock<std::mutex> lk(m);
    task = &fn;
    pending = (int)kids.size();
    std::fill(rc.begin(), rc.end(), 0);
    epoch++;
    cv_task.notify_all();
    cv_done.wait(lk, [&] { return pending == 0; });
    task = nullptr;
    for (int v : rc)
      if (v) return v;
    return ATMRT_OK;
  }

  void worker(int i) {
    (void)hipSetDevice(devices[i]);
    uint64_t seen = 0;
    for (;;) {
      const std::function<int(atmrt_ctx*, int)>* fn;
      {
        std::unique_lock<std::mutex> lk(m);
        cv_task.wait(lk, [&] { return stop || epoch != seen; });
        if (stop) return;
        seen = epoch;
        fn = task;
      }
      const int r = (*fn)(kids[i], i);
      if (r) abort_task(); // peers that wait for this device at a barrier of the same task give up instead of hanging
      {
        std::lock_guard<std::mutex> lk(m);
        rc[i] = r;
        if (--pending == 0) cv_done.notify_all();
      }
    }
  }
};

// Equal tiles, their inner boundaries on multiples of 64 columns when every tile is at least 128 wide (tiles_rebalance below says
// why: a wavefront then is 64 adjacent columns of one row; a 500-column tile marches 15 % slower than a 512-column one, which is
// more than the 64 columns of imbalance the alignment can cost).  Widths that divide evenly into multiples of 64 — 4096 or 8192
// over 2, 4, 8 ranks — are cut exactly as g W / G.
static int tile_begin_aligned(int width, int g, int world) {
  const int b = shard_begin(width, g, world);
  if (g == 0 || g == world || width < 128 * world) return b;
  return (b + 32) / 64 * 64;
}

// The tiling the next frame of a `width`-column image uses: equal widths whenever the width (or nothing yet) says so.
static void comm_tiling(Comm* cm, int width) {
  const int G = cm->world;
  if (cm->cols_W == width && (int)cm->cols.size() == G + 1) return;
  cm->cols.resize((size_t)G + 1);
  for (int g = 0; g <= G; g++) cm->cols[(size_t)g] = tile_begin_aligned(width, g, G);
  cm->cols_W = width;
  cm->cols_pinned = false;
}

void comm_columns(const atmrt_ctx* c, int width, int* c0, int* c1) {
  if (!c->comm) return;
  comm_tiling(c->comm, width);
  *c0 = c->comm->cols[(size_t)c->comm->rank];
  *c1 = c->comm->cols[(size_t)c->comm->rank + 1];
}

// Tile boundaries that would have equalised the tiles' times, had the cost per column been constant inside each tile: the inverse
// of the piecewise-linear cumulative cost at k / G of its total — moved to the nearest multiple of 64 columns.  A wavefront of the
// marching kernels is 64 consecutive pixels of the tile's row-major order: with a width that is a multiple of 64 it is 64 adjacent
// columns of ONE row (rays that leave the terrain's height range and end together); with any other width most wavefronts straddle
// two rows, i.e. both edges of the tile, and the march of the headline's tiles takes 15 % longer (measured, round 4:
// profiles/r04/shard_balance_unquantised_recut.json — 8 x 30.2 ms with 512-column tiles, 8 x 34.8 ms with 490 .. 559).  Images
// narrower than 128 columns per tile are cut to the column (nothing to keep aligned).  The re-cut is adopted only if, under the
// same model, it shortens the slowest tile by at least 1 %.  A pure function of its arguments — every rank evaluates it on the same
// gathered numbers and gets the same tiling.  Widths stay >= 1.  Returns false (and copies the input) when a time is not a positive
// finite number or nothing is to be gained.
// returns 1: a new cut, 0: the tiling stays (nothing to gain), -1: invalid input
int tiles_rebalance(int W, int G, const int* cols, const double* ms, int* out) {
  for (int g = 0; g <= G; g++) out[g] = cols[g];
  double total = 0.0, worst = 0.0;
  for (int g = 0; g < G; g++) {
    if (!(ms[g] > 0.0) || !(ms[g] < 1e300) || cols[g + 1] <= cols[g]) return -1;
    total += ms[g];
    worst = std::max(worst, ms[g]);
  }
  if (cols[0] != 0 || cols[G] != W) return -1;
  const int q = W >= 128 * G ? 64 : 1; // the quantum of a boundary
  std::vector<int> next((size_t)G + 1);
  next[0] = 0, next[(size_t)G] = W;
  int g = 0;
  double before = 0.0; // cost of the tiles left of tile g
  for (int k = 1; k < G; k++) {
    const double want = total * (double)k / (double)G;
    while (g < G - 1 && before + ms[g] < want) before += ms[g], g++;
    const double x = (double)cols[g] + (want - before) / ms[g] * (double)(cols[g + 1] - cols[g]);
    next[(size_t)k] = (int)(x / q + 0.5) * q;
  }
  for (int k = 1; k < G; k++) next[(size_t)k] = std::max(next[(size_t)k], next[(size_t)k - 1] + q);
  for (int k = G - 1; k >= 1; k--) next[(size_t)k] = std::min(next[(size_t)k], next[(size_t)k + 1] - (k + 1 == G ? 1 : q));
  for (int k = 1; k <= G; k++)
    if (next[(size_t)k] <= next[(size_t)k - 1]) return 0; // (cannot happen for W >= G; guards the arithmetic above)
  // the slowest tile of the new cut under the model: cost of [a, b) = sum over old tiles of their density x overlap
  double new_worst = 0.0;
  for (int k = 0; k < G; k++) {
    double cost = 0.0;
    for (int j = 0; j < G; j++) {
      const int lo = std::max(next[(size_t)k], cols[j]), hi = std::min(next[(size_t)k + 1], cols[j + 1]);
      if (hi > lo) cost += ms[j] * (double)(hi - lo) / (double)(cols[j + 1] - cols[j]);
    }
    new_worst = std::max(new_worst, cost);
  }
  if (!(new_worst <= 0.99 * worst)) return 0;
  for (int k = 0; k <= G; k++) out[k] = next[(size_t)k];
  return 1;
}

int multi_size(const atmrt_ctx* parent) { return parent->multi ? (int)parent->multi->kids.size() : 1; }
atmrt_ctx* multi_child(atmrt_ctx* parent, int i) { return parent->multi->kids[(size_t)i]; }

int multi_forward(atmrt_ctx* parent, const std::function<int(atmrt_ctx*)>& fn) {
  MultiGroup* g = parent->multi;
  int rc = g->run([&](atmrt_ctx* k, int) { return fn(k); });
  if (rc) {
    // the message of the device that failed FIRST-HAND, not of one that was released from a barrier because of it
    int who = -1;
    for (size_t i = 0; i < g->kids.size(); i++) {
      if (!g->rc[i]) continue;
      if (who < 0) who = (int)i;
      if (g->kids[i]->error.find("another device of the context failed") == std::string::npos) {
        who = (int)i;
        break;
      }
    }
    rc = g->rc[(size_t)who];
    parent->error = "device " + std::to_string(g->devices[(size_t)who]) + " (tile " + std::to_string(who) + "): " + g->kids[(size_t)who]->error;
  }
  return rc;
}

} // namespace atmrt

// ---------------------------------------------------------------------------------------------
// kernels: rank-major tiles -> the row-major image
// ---------------------------------------------------------------------------------------------
namespace {

// the tile of column x: the last g with cols[g] <= x (cols: world + 1 ascending boundaries, cols[0] = 0, cols[world] = W)
__device__ __forceinline__ int tile_of_column(int x, const int* __restrict__ cols, int world) {
  int lo = 0, hi = world;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (cols[mid] <= x) lo = mid;
    else hi = mid;
  }
  return lo;
}

// gathered: [G] slabs of `slab_bytes`; the slab of tile g is exactly what the generators write for a tile of H x wl_g pixels: 10 f64
// planes [H][wl_g] back to back (the planar normal is the last three) and the u32 hit_count plane.  One thread per image pixel,
// lanes = adjacent columns: reads and writes are coalesced inside a tile.
__global__ __launch_bounds__(256) void k_assemble_image(const char* __restrict__ gathered, size_t slab_bytes, int W, int H, int G,
                                                         const int* __restrict__ cols, DensePlanes image) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t npx = (size_t)W * H;
  if (p >= npx) return;
  const int y = (int)(p / (size_t)W), x = (int)(p % (size_t)W);
  const int g = tile_of_column(x, cols, G);
  const int c0 = cols[g], wl = cols[g + 1] - c0;
  const size_t si = (size_t)y * wl + (size_t)(x - c0), plane_stride = (size_t)H * wl;
  const double* src = reinterpret_cast<const double*>(gathered + (size_t)g * slab_bytes);
  image.azimuth[p] = src[0 * plane_stride + si];
  image.elevation_angle[p] = src[1 * plane_stride + si];
  image.lat[p] = src[2 * plane_stride + si];
  image.lon[p] = src[3 * plane_stride + si];
  image.distance[p] = src[4 * plane_stride + si];
  image.elevation[p] = src[5 * plane_stride + si];
  image.path_length[p] = src[6 * plane_stride + si];
  image.normal[p] = src[7 * plane_stride + si];
  image.normal[npx + p] = src[8 * plane_stride + si];
  image.normal[2 * npx + p] = src[9 * plane_stride + si];
  image.hit_count[p] = reinterpret_cast<const uint32_t*>(src + N_F64_PLANES * plane_stride)[si];
}

__global__ __launch_bounds__(256) void k_assemble_rgb(const uint8_t* __restrict__ gathered, size_t tile_bytes, int W, int H, int G,
                                                       const int* __restrict__ cols, uint8_t* __restrict__ rgb) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (size_t)W * H) return;
  const int y = (int)(p / (size_t)W), x = (int)(p % (size_t)W);
  const int g = tile_of_column(x, cols, G);
  const int c0 = cols[g], wl = cols[g + 1] - c0;
  const uint8_t* s = gathered + (size_t)g * tile_bytes + 3 * ((size_t)y * wl + (size_t)(x - c0));
  rgb[3 * p + 0] = s[0];
  rgb[3 * p + 1] = s[1];
  rgb[3 * p + 2] = s[2];
}

// The packed lists of one rank, laid out for a capacity of n entries (the largest rank's count): what is all-gathered.
struct HitBlock {
  __host__ __device__ static size_t bytes(size_t n) { return (n * (12 * 8 + 4) + 255) / 256 * 256; }
  __host__ __device__ static size_t off_f64(int field, size_t n) { return (size_t)field * n * 8; } // lat 0 lon 1 distance 2 elevation 3 path_length 4 normal 5 rgba 8
  __host__ __device__ static size_t off_tag(size_t n) { return 12 * n * 8; }
};

// One thread per image pixel: its trace points move from its rank's block (at the rank's own offsets) to the image's offsets.
__global__ __launch_bounds__(256) void k_gather_hits(const char* __restrict__ blocks, size_t block_bytes, size_t n_cap,
                                                      const uint64_t* __restrict__ loc_off, size_t off_stride, int W, int H, int G,
                                                      const int* __restrict__ cols, const uint32_t* __restrict__ hit_count,
                                                      const uint64_t* __restrict__ img_off, PackedHits out) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (size_t)W * H) return;
  const uint32_t cnt = hit_count[p];
  if (!cnt) return;
  const int y = (int)(p / (size_t)W), x = (int)(p % (size_t)W);
  const int g = tile_of_column(x, cols, G);
  const int c0 = cols[g], wl = cols[g + 1] - c0;
  const uint64_t s0 = loc_off[(size_t)g * off_stride + (size_t)y * wl + (size_t)(x - c0)];
  const uint64_t d0 = img_off[p];
  const char* blk = blocks + (size_t)g * block_bytes;
  const double* lat = reinterpret_cast<const double*>(blk + HitBlock::off_f64(0, n_cap));
  const double* lon = reinterpret_cast<const double*>(blk + HitBlock::off_f64(1, n_cap));
  const double* dist = reinterpret_cast<const double*>(blk + HitBlock::off_f64(2, n_cap));
  const double* elev = reinterpret_cast<const double*>(blk + HitBlock::off_f64(3, n_cap));
  const double* plen = reinterpret_cast<const double*>(blk + HitBlock::off_f64(4, n_cap));
  const double* nrm = reinterpret_cast<const double*>(blk + HitBlock::off_f64(5, n_cap));
  const double* rgba = reinterpret_cast<const double*>(blk + HitBlock::off_f64(8, n_cap));
  const uint32_t* tag = reinterpret_cast<const uint32_t*>(blk + HitBlock::off_tag(n_cap));
  for (uint32_t j = 0; j < cnt; j++) {
    const uint64_t s = s0 + j, d = d0 + j;
    out.lat[d] = lat[s];
    out.lon[d] = lon[s];
    out.distance[d] = dist[s];
    out.elevation[d] = elev[s];
    out.path_length[d] = plen[s];
    for (int k = 0; k < 3; k++) out.normal[3 * d + k] = nrm[3 * s + k];
    for (int k = 0; k < 4; k++) out.rgba[4 * d + k] = rgba[4 * s + k];
    out.color_tag[d] = tag[s];
  }
}

unsigned blocks_for(size_t n) { return (unsigned)((n + 255) / 256); }

// ---------------------------------------------------------------------------------------------
// the exchange
// ---------------------------------------------------------------------------------------------
#define NCCL_TRY(ctx, expr)                                                                                      \
  do {                                                                                                           \
    ncclResult_t r_ = (expr);                                                                                    \
    if (r_ != ncclSuccess) return (ctx)->fail(ATMRT_ERR_HIP, "%s failed: %s", #expr, rccl()->GetErrorString(r_)); \
  } while (0)

// A group context (several devices of ONE process) on the RCCL route: every device's host thread passes here before it enqueues
// its part of a collective.  (1) Nothing that may allocate or map device memory runs on one thread while another device of the
// process already sits inside the collective's kernel waiting for it — hipMalloc with peer access enabled may wait for the peers.
// (2) A device that has failed (or where a failure was injected) releases the others here instead of leaving them inside
// ncclAllGather, which has no way out.  The peer-copy route has its own two barriers.
static int group_gate(atmrt_ctx* c) {
  Comm* cm = c->comm;
  if (!cm->group || cm->world == 1) return ATMRT_OK;
  if (!cm->group->barrier()) return c->fail(ATMRT_ERR_STATE, "another device of the context failed during this frame");
  return ATMRT_OK;
}

// Every rank's `bytes` at `send` -> all of them, rank-major, at `recv` (both in this rank's HBM), ordered on c->stream.
int comm_all_gather(atmrt_ctx* c, const void* send, void* recv, size_t bytes) {
  Comm* cm = c->comm;
  hipStream_t s = c->stream;
  if (cm && cm->fail_countdown > 0 && --cm->fail_countdown == 0) { // atmrt_debug_fail_next_collective
    if (cm->group) cm->group->abort_task();
    return c->fail(ATMRT_ERR_HIP, "collective failure injected by atmrt_debug_fail_next_collective (rank %d)", cm->rank);
  }
  if (!cm || (cm->world == 1 && cm->route != ATMRT_ROUTE_EXTERNAL && cm->route != ATMRT_ROUTE_EXTERNAL_DEVICE)) {
    if (cm && cm->nccl) { // RCCL at world size 1: the same call as with 8 ranks
      NCCL_TRY(c, rccl()->AllGather(send, recv, bytes, ncclUint8, cm->nccl, s));
      return ATMRT_OK;
    }
    HIP_TRY(c, hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s));
    return ATMRT_OK;
  }
  switch (cm->route) {
    case ATMRT_ROUTE_RCCL: {
      int rc = group_gate(c);
      if (rc) return rc;
      ncclResult_t nr = rccl()->AllGather(send, recv, bytes, ncclUint8, cm->nccl, s);
      if (nr != ncclSuccess) {
        if (cm->group) cm->group->abort_task();
        return c->fail(ATMRT_ERR_HIP, "ncclAllGather of %zu bytes per rank failed on rank %d: %s", bytes, cm->rank, rccl()->GetErrorString(nr));
      }
      return ATMRT_OK;
    }
    case ATMRT_ROUTE_PEER: {
      // one process: every device writes its tile into every peer's buffer.  A failing rank still passes both barriers.
      MultiGroup* g = cm->group;
      g->recv_ptr[(size_t)cm->rank] = recv;
      if (!g->barrier()) // every buffer is published, and no peer still reads what is about to be overwritten
        return c->fail(ATMRT_ERR_STATE, "another device of the context failed during this frame");
      hipError_t e = hipSuccess;
      for (int q = 0; q < cm->world && e == hipSuccess; q++)
        e = hipMemcpyPeerAsync(static_cast<char*>(g->recv_ptr[(size_t)q]) + (size_t)cm->rank * bytes, g->devices[(size_t)q], send,
                               c->device, bytes, s);
      if (e == hipSuccess) e = hipStreamSynchronize(s);
      if (e != hipSuccess) {
        g->abort_task();
        return c->fail(ATMRT_ERR_HIP, "peer copy of a tile failed: %s", hipGetErrorString(e));
      }
      if (!g->barrier()) // every tile has landed everywhere
        return c->fail(ATMRT_ERR_STATE, "another device of the context failed during this frame");
      return ATMRT_OK;
    }
    case ATMRT_ROUTE_EXTERNAL: {
      HIP_TRY(c, cm->h_send.reserve(bytes));
      HIP_TRY(c, cm->h_recv.reserve(bytes * (size_t)cm->world));
      HIP_TRY(c, hipMemcpyAsync(cm->h_send.ptr, send, bytes, hipMemcpyDeviceToHost, s));
      HIP_TRY(c, hipStreamSynchronize(s));
      const int rc = cm->ext(cm->ext_user, cm->h_send.ptr, cm->h_recv.ptr, bytes);
      if (rc) return c->fail(ATMRT_ERR_HIP, "the host's all-gather callback returned %d", rc);
      HIP_TRY(c, hipMemcpyAsync(recv, cm->h_recv.ptr, bytes * (size_t)cm->world, hipMemcpyHostToDevice, s));
      return ATMRT_OK;
    }
    case ATMRT_ROUTE_EXTERNAL_DEVICE: {
      HIP_TRY(c, hipStreamSynchronize(s));
      const int rc = cm->ext_dev(cm->ext_user, send, recv, bytes);
      if (rc) return c->fail(ATMRT_ERR_HIP, "the host's device all-gather callback returned %d", rc);
      return ATMRT_OK;
    }
    default:
      return c->fail(ATMRT_ERR_STATE, "this context has no transport for %d ranks", cm->world);
  }
}

int ensure_comm_events(atmrt_ctx* c) {
  Comm* cm = c->comm;
  if (cm->ev_g0) return ATMRT_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, hipEventCreate(&cm->ev_g0));
  HIP_TRY(c, hipEventCreate(&cm->ev_g1));
  HIP_TRY(c, hipEventCreate(&cm->ev_a1));
  return ATMRT_OK;
}

// The first collective of a communicator, run at set-up time where a host can still choose another transport: every rank sends
// its rank number, every rank must find 0 .. world - 1 in order.  RCCL has no time-out of its own, so the wait is a poll with one
// (ATMRT_COMM_PROBE_TIMEOUT seconds, default 180: the first collective of an 8-rank communicator builds its rings and can take
// tens of seconds); a communicator that timed out is aborted (ncclCommAbort) and given up.
int comm_probe(atmrt_ctx* c) {
  Comm* cm = c->comm;
  Rccl* r = rccl();
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, cm->d_small.reserve(512 + 8 * (size_t)cm->world));
  uint64_t* d_mine = cm->d_small.as<uint64_t>();
  uint64_t* d_all = d_mine + 64;
  const uint64_t mine = 0xa7e0000000000000ull | (uint64_t)cm->rank;
  hipStream_t s = c->stream;
  HIP_TRY(c, hipMemcpyAsync(d_mine, &mine, 8, hipMemcpyHostToDevice, s));
  HIP_TRY(c, hipMemsetAsync(d_all, 0, 8 * (size_t)cm->world, s));
  int rc = group_gate(c);
  if (rc) return rc;
  ncclResult_t nr = r->AllGather(d_mine, d_all, 8, ncclUint8, cm->nccl, s);
  if (nr != ncclSuccess) {
    if (cm->group) cm->group->abort_task();
    return c->fail(ATMRT_ERR_HIP, "the probe ncclAllGather failed on rank %d: %s", cm->rank, r->GetErrorString(nr));
  }
  const char* env = getenv("ATMRT_COMM_PROBE_TIMEOUT");
  const double limit = env && atof(env) > 0.0 ? atof(env) : 180.0;
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    const hipError_t q = hipStreamQuery(s);
    if (q == hipSuccess) break;
    if (q != hipErrorNotReady) return c->fail(ATMRT_ERR_HIP, "the probe collective failed on rank %d: %s", cm->rank, hipGetErrorString(q));
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > limit) {
      if (r->CommAbort) (void)r->CommAbort(cm->nccl);
      cm->nccl = nullptr; // aborted (or, without ncclCommAbort, abandoned): never touched again
      if (cm->group && (size_t)cm->rank < cm->group->nccl.size()) cm->group->nccl[(size_t)cm->rank] = nullptr;
      if (cm->group) cm->group->abort_task();
      return c->fail(ATMRT_ERR_HIP, "the probe collective of rank %d did not complete within %.0f s (ATMRT_COMM_PROBE_TIMEOUT)", cm->rank, limit);
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(1));
  }
  std::vector<uint64_t> all((size_t)cm->world);
  HIP_TRY(c, hipMemcpy(all.data(), d_all, 8 * (size_t)cm->world, hipMemcpyDeviceToHost));
  for (int g = 0; g < cm->world; g++)
    if (all[(size_t)g] != (0xa7e0000000000000ull | (uint64_t)g))
      return c->fail(ATMRT_ERR_HIP, "the probe collective delivered %016llx where rank %d's word belongs", (unsigned long long)all[(size_t)g], g);
  return ATMRT_OK;
}

// geometry of the tiles of the frame the context is about to generate
int frame_geometry(atmrt_ctx* c) {
  Comm* cm = c->comm;
  if (!c->have_params) return c->fail(ATMRT_ERR_STATE, "atmrt_set_params has not been called");
  const int W = c->params.width, H = c->params.height, G = cm->world;
  if (W < G) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "image width %d is less than the %d ranks: a rank would be left without a column", W, G);
  comm_tiling(cm, W);
  cm->cols_frame = cm->cols;
  int wl_max = 0;
  for (int g = 0; g < G; g++) wl_max = std::max(wl_max, cm->cols[(size_t)g + 1] - cm->cols[(size_t)g]);
  cm->W = W, cm->H = H, cm->wl_max = wl_max;
  cm->slab_bytes = pad256((size_t)H * wl_max * (N_F64_PLANES * 8 + 4)) + SLAB_TRAILER_BYTES;
  return ATMRT_OK;
}

// Phase A of a shared frame: this rank's tile into its slab — the nine planes the generators write, back to back in ONE buffer
// (84 B per pixel), so that one collective moves them all.  Every buffer the frame's exchange will need is reserved HERE, before
// any rank can be inside a collective (group_gate).
int tile_generate(atmrt_ctx* c, uint64_t* ray_steps, double* device_ms) {
  int rc = frame_geometry(c);
  if (rc) return rc;
  if ((rc = ensure_comm_events(c))) return rc;
  Comm* cm = c->comm;
  cm->image_valid = false;
  cm->exchanged = false;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, cm->d_slab.reserve(cm->slab_bytes));
  HIP_TRY(c, cm->d_gathered.reserve(cm->slab_bytes * (size_t)cm->world));
  HIP_TRY(c, cm->d_tiling.reserve(4 * ((size_t)cm->world + 1)));
  HIP_TRY(c, cm->d_small.reserve(512 + 8 * (size_t)cm->world));
  const int c0 = cm->cols_frame[(size_t)cm->rank], c1 = cm->cols_frame[(size_t)cm->rank + 1];
  const size_t ps = (size_t)cm->H * (size_t)(c1 - c0);
  double* base = cm->d_slab.as<double>();
  DensePlanes& d = cm->slab_planes;
  d.azimuth = base, d.elevation_angle = base + ps, d.lat = base + 2 * ps, d.lon = base + 3 * ps, d.distance = base + 4 * ps;
  d.elevation = base + 5 * ps, d.path_length = base + 6 * ps, d.normal = base + 7 * ps; // planar [3][H][wl]
  d.hit_count = reinterpret_cast<uint32_t*>(base + N_F64_PLANES * ps);
  uint64_t nh = 0;
  return api_generate_tile(c, &cm->slab_planes, false, &nh, ray_steps, device_ms);
}

// After a frame: the tiling of the next one, from every tile's time (the same numbers on every rank).
void tile_rebalance(Comm* cm) {
  if (!cm->balance || cm->cols_pinned || cm->world < 2) return;
  const int G = cm->world;
  std::vector<double> ms((size_t)G);
  double sum = 0.0, worst = 0.0;
  for (int g = 0; g < G; g++) {
    ms[(size_t)g] = cm->trailers[(size_t)g].tile_ms;
    sum += ms[(size_t)g];
    worst = std::max(worst, ms[(size_t)g]);
  }
  if (!(sum > 0.0) || !(worst * G > 1.01 * sum)) return; // balanced within 1 %: leave it (the times carry that much noise)
  std::vector<int> next((size_t)G + 1);
  if (tiles_rebalance(cm->W, G, cm->cols_frame.data(), ms.data(), next.data()) == 1) cm->cols = next;
}

// Phase B: the collective + the permutation into `image` (planes on this rank's device; NULL azimuth: take part, assemble nothing).
int tile_exchange(atmrt_ctx* c, const atmrt_device_planes_t* image) {
  Comm* cm = c->comm;
  hipStream_t s = c->stream;
  const int G = cm->world;
  HIP_TRY(c, hipSetDevice(c->device));
  SlabTrailer mine{};
  mine.n_hits = c->last_packed ? c->last_nhits : 0;
  mine.seq = cm->seq;
  mine.tile_ms = c->timings.total_ms;
  mine.packed = c->last_packed ? 1u : 0u;
  mine.c0 = (uint32_t)cm->cols_frame[(size_t)cm->rank];
  mine.c1 = (uint32_t)cm->cols_frame[(size_t)cm->rank + 1];
  const size_t trailer_at = cm->slab_bytes - SLAB_TRAILER_BYTES;
  HIP_TRY(c, hipMemcpyAsync(cm->d_slab.as<char>() + trailer_at, &mine, sizeof mine, hipMemcpyHostToDevice, s));
  HIP_TRY(c, hipMemcpyAsync(cm->d_tiling.ptr, cm->cols_frame.data(), 4 * ((size_t)G + 1), hipMemcpyHostToDevice, s));
  HIP_TRY(c, hipStreamSynchronize(s)); // both sources are host stack / vectors: done with them before anything else happens
  HIP_TRY(c, hipEventRecord(cm->ev_g0, s));
  int rc = comm_all_gather(c, cm->d_slab.ptr, cm->d_gathered.ptr, cm->slab_bytes);
  if (rc) return rc;
  HIP_TRY(c, hipEventRecord(cm->ev_g1, s));
  if (image && image->azimuth) {
    DensePlanes img;
    img.azimuth = image->azimuth, img.elevation_angle = image->elevation_angle, img.hit_count = image->hit_count;
    img.lat = image->lat, img.lon = image->lon, img.distance = image->distance, img.elevation = image->elevation;
    img.path_length = image->path_length, img.normal = image->normal;
    hipLaunchKernelGGL(k_assemble_image, dim3(blocks_for((size_t)cm->W * cm->H)), dim3(256), 0, s, cm->d_gathered.as<char>(),
                       cm->slab_bytes, cm->W, cm->H, G, (const int*)cm->d_tiling.as<int>(), img);
    cm->last_image = img;
  }
  HIP_TRY(c, hipEventRecord(cm->ev_a1, s));
  cm->trailers.assign((size_t)G, SlabTrailer{});
  HIP_TRY(c, hipMemcpy2DAsync(cm->trailers.data(), sizeof(SlabTrailer), cm->d_gathered.as<char>() + trailer_at, cm->slab_bytes,
                              sizeof(SlabTrailer), (size_t)G, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  HIP_TRY(c, hipGetLastError());
  // every rank must have contributed the same frame, cut the same way: anything else is a host that lost step (a rank that
  // skipped a frame, changed its parameters alone) and would otherwise be a silently mixed image
  uint64_t seq_max = 0;
  for (const SlabTrailer& t : cm->trailers) seq_max = std::max(seq_max, t.seq);
  for (int g = 0; g < G; g++) {
    const SlabTrailer& t = cm->trailers[(size_t)g];
    if (t.seq != mine.seq || t.packed != mine.packed || (int)t.c0 != cm->cols_frame[(size_t)g] || (int)t.c1 != cm->cols_frame[(size_t)g + 1]) {
      cm->seq = seq_max + 1; // every rank sees the same trailers and takes the same number: the next frame is in step again
      cm->cols_W = -1;       // and cut into equal tiles
      return c->fail(ATMRT_ERR_STATE, "rank %d contributed frame %llu, columns [%u, %u), lists %u where rank %d has frame %llu, columns [%d, %d), "
                                      "lists %u: the ranks were out of step (this frame is void, the next one is in step again)", g,
                     (unsigned long long)t.seq, t.c0, t.c1, t.packed, cm->rank, (unsigned long long)mine.seq, cm->cols_frame[(size_t)g],
                     cm->cols_frame[(size_t)g + 1], mine.packed);
    }
  }
  cm->seq++;
  cm->exchanged = true;
  cm->image_valid = image && image->azimuth;
  float g_ms = 0.f, a_ms = 0.f;
  HIP_TRY(c, hipEventElapsedTime(&g_ms, cm->ev_g0, cm->ev_g1));
  HIP_TRY(c, hipEventElapsedTime(&a_ms, cm->ev_g1, cm->ev_a1));
  cm->tm = atmrt_comm_timings_t{};
  cm->tm.gather_ms = g_ms;
  cm->tm.assemble_ms = a_ms;
  cm->tm.tile_ms_max = 0.0, cm->tm.tile_ms_min = 1e300;
  for (const SlabTrailer& t : cm->trailers) {
    cm->tm.tile_ms_max = std::max(cm->tm.tile_ms_max, t.tile_ms);
    cm->tm.tile_ms_min = std::min(cm->tm.tile_ms_min, t.tile_ms);
  }
  cm->tm.bytes_per_rank = cm->slab_bytes;
  cm->tm.world = G;
  cm->tm.route = cm->nccl ? ATMRT_ROUTE_RCCL : cm->route;
  cm->tm.collectives = 1;
  tile_rebalance(cm);
  return ATMRT_OK;
}

bool image_planes_complete(const atmrt_device_planes_t& p) {
  return p.azimuth && p.elevation_angle && p.hit_count && p.lat && p.lon && p.distance && p.elevation && p.path_length && p.normal;
}

// The lists of the last shared frame on this rank.  dst == NULL: the image's total only — known to every rank since the frame's own
// collective (SlabTrailer), no communication.  dst != NULL: ONE collective, the all-gather of the packed lists, in which every
// rank takes part whatever it wants for itself (dst->hit_offset == NULL: nothing; an error of its own — capacity too small, a missing
// array, no image planes on this rank — is reported AFTER the collective): the ranks can never disagree about whether it happens,
// because that depends only on numbers all of them hold — an image without a single trace point has no collective at all.
int tile_hits(atmrt_ctx* c, const atmrt_device_hits_t* dst, uint64_t* n_total_out) {
  Comm* cm = c->comm;
  hipStream_t s = c->stream;
  if (!cm->exchanged || !c->last_valid)
    return c->fail(ATMRT_ERR_STATE, "atmrt_image_hits_device needs a frame: call atmrt_generate_image_device first");
  if (!c->last_packed)
    return c->fail(ATMRT_ERR_STATE, "the last frame holds first-hit planes only (opaque scene): its trace points are the planes themselves");
  const int G = cm->world;
  uint64_t n_total = 0, n_cap = 1;
  for (const SlabTrailer& t : cm->trailers) n_total += t.n_hits, n_cap = std::max<uint64_t>(n_cap, t.n_hits);
  if (n_total_out) *n_total_out = n_total;
  if (!dst) return ATMRT_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  const bool wants = dst->hit_offset != nullptr;
  const size_t npx = (size_t)cm->W * cm->H;
  if (n_total == 0) { // every rank knows: no collective.  The offsets of an image without trace points are all zero.
    if (wants) {
      HIP_TRY(c, hipMemsetAsync(dst->hit_offset, 0, npx * 8, s));
      HIP_TRY(c, hipStreamSynchronize(s));
    }
    return ATMRT_OK;
  }
  // this rank's own troubles, reported after it has done its part
  const char* trouble = nullptr;
  if (wants) {
    if (!cm->image_valid) trouble = "the image planes of the last frame were not assembled on this device";
    else if (dst->capacity < n_total) trouble = "capacity is less than the trace points of the image";
    else if (!dst->lat || !dst->lon || !dst->distance || !dst->elevation || !dst->path_length || !dst->normal || !dst->color_tag || !dst->rgba)
      trouble = "every array pointer must be a device allocation";
  }
  // (1) this rank's lists into a block laid out for the largest rank, (2) one all-gather of the blocks
  const size_t bb = HitBlock::bytes(n_cap);
  const size_t ps = (size_t)cm->H * cm->wl_max; // stride of the per-rank offset tables
  HIP_TRY(c, cm->d_hits_send.reserve(bb));
  HIP_TRY(c, cm->d_hits_recv.reserve(bb * (size_t)G));
  HIP_TRY(c, cm->d_loc_off.reserve(ps * (size_t)G * 8));
  HIP_TRY(c, cm->d_scan_tmp.reserve((std::max(npx, ps) / 2048 + 4) * 8 + 64));
  char* blk = cm->d_hits_send.as<char>();
  const PackedHits& h = c->last_hits;
  const uint64_t n_local = c->last_nhits;
  auto d2d = [&](size_t off, const void* from, size_t bytes) {
    return bytes ? hipMemcpyAsync(blk + off, from, bytes, hipMemcpyDeviceToDevice, s) : hipSuccess;
  };
  HIP_TRY(c, d2d(HitBlock::off_f64(0, n_cap), h.lat, n_local * 8));
  HIP_TRY(c, d2d(HitBlock::off_f64(1, n_cap), h.lon, n_local * 8));
  HIP_TRY(c, d2d(HitBlock::off_f64(2, n_cap), h.distance, n_local * 8));
  HIP_TRY(c, d2d(HitBlock::off_f64(3, n_cap), h.elevation, n_local * 8));
  HIP_TRY(c, d2d(HitBlock::off_f64(4, n_cap), h.path_length, n_local * 8));
  HIP_TRY(c, d2d(HitBlock::off_f64(5, n_cap), h.normal, n_local * 24));
  HIP_TRY(c, d2d(HitBlock::off_f64(8, n_cap), h.rgba, n_local * 32));
  HIP_TRY(c, d2d(HitBlock::off_tag(n_cap), h.color_tag, n_local * 4));
  int rc = comm_all_gather(c, blk, cm->d_hits_recv.ptr, bb);
  if (rc) return rc;
  cm->tm.collectives = 2;
  if (wants && !trouble) {
    // (3) offsets: every rank's own (a scan of its hit_count plane inside the gathered slabs) and the image's
    uint64_t* tmp = cm->d_scan_tmp.as<uint64_t>();
    unsigned long long* total = reinterpret_cast<unsigned long long*>(cm->d_small.as<uint64_t>() + 8); // scratch: the scans' grand totals are not needed
    for (int g = 0; g < G; g++) {
      const size_t n = (size_t)cm->H * (size_t)(cm->cols_frame[(size_t)g + 1] - cm->cols_frame[(size_t)g]);
      const uint32_t* counts = reinterpret_cast<const uint32_t*>(cm->d_gathered.as<char>() + (size_t)g * cm->slab_bytes + N_F64_PLANES * n * 8);
      launch_scan_u32(counts, n, tmp, cm->d_loc_off.as<uint64_t>() + (size_t)g * ps, total, s);
    }
    launch_scan_u32(cm->last_image.hit_count, npx, tmp, dst->hit_offset, total, s);
    PackedHits out;
    out.lat = dst->lat, out.lon = dst->lon, out.distance = dst->distance, out.elevation = dst->elevation;
    out.path_length = dst->path_length, out.normal = dst->normal, out.color_tag = dst->color_tag, out.rgba = dst->rgba;
    hipLaunchKernelGGL(k_gather_hits, dim3(blocks_for(npx)), dim3(256), 0, s, cm->d_hits_recv.as<char>(), bb, (size_t)n_cap,
                       cm->d_loc_off.as<uint64_t>(), ps, cm->W, cm->H, G, (const int*)cm->d_tiling.as<int>(),
                       (const uint32_t*)cm->last_image.hit_count, (const uint64_t*)dst->hit_offset, out);
  }
  HIP_TRY(c, hipStreamSynchronize(s));
  HIP_TRY(c, hipGetLastError());
  if (trouble)
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "%s (the image holds %llu trace points, capacity %llu)", trouble, (unsigned long long)n_total,
                   (unsigned long long)dst->capacity);
  return ATMRT_OK;
}

// renderer::draw_image of this rank's tile + the 3 B/pixel exchange
int tile_draw(atmrt_ctx* c, const atmrt_coloring_t* coloring, uint8_t* rgb_image) {
  Comm* cm = c->comm;
  hipStream_t s = c->stream;
  if (!cm->exchanged || !c->last_valid)
    return c->fail(ATMRT_ERR_STATE, "atmrt_draw_image_gathered_device needs a frame: call atmrt_generate_image_device first");
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t tile_bytes = pad256(3 * (size_t)cm->H * cm->wl_max);
  HIP_TRY(c, cm->d_rgb_tile.reserve(tile_bytes));
  HIP_TRY(c, cm->d_rgb_all.reserve(tile_bytes * (size_t)cm->world));
  int rc = atmrt_draw_image_device(c, coloring, cm->d_rgb_tile.as<uint8_t>());
  if (rc) {
    if (cm->group) cm->group->abort_task();
    return rc;
  }
  if ((rc = comm_all_gather(c, cm->d_rgb_tile.ptr, cm->d_rgb_all.ptr, tile_bytes))) return rc;
  if (rgb_image)
    hipLaunchKernelGGL(k_assemble_rgb, dim3(blocks_for((size_t)cm->W * cm->H)), dim3(256), 0, s, cm->d_rgb_all.as<uint8_t>(), tile_bytes,
                       cm->W, cm->H, cm->world, (const int*)cm->d_tiling.as<int>(), rgb_image);
  HIP_TRY(c, hipStreamSynchronize(s));
  HIP_TRY(c, hipGetLastError());
  return ATMRT_OK;
}

int comm_attach(atmrt_ctx* c, int rank, int world) {
  if (c->multi) return c->fail(ATMRT_ERR_STATE, "a multi-device context already shares its frames among its own devices");
  if (c->comm) return c->fail(ATMRT_ERR_STATE, "this context already belongs to a group of ranks");
  if (world < 1 || rank < 0 || rank >= world || world > 4096) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "rank %d outside world %d", rank, world);
  c->comm = new Comm();
  c->comm->rank = rank;
  c->comm->world = world;
  return ATMRT_OK;
}

} // namespace

void atmrt::comm_destroy(atmrt_ctx* c) {
  Comm* cm = c->comm;
  if (!cm) return;
  (void)hipSetDevice(c->device);
  if (cm->nccl && !cm->group) (void)rccl()->CommDestroy(cm->nccl); // a group destroys the communicators it created
  for (hipEvent_t ev : {cm->ev_g0, cm->ev_g1, cm->ev_a1})
    if (ev) (void)hipEventDestroy(ev);
  delete cm;
  c->comm = nullptr;
}

void atmrt::multi_destroy(atmrt_ctx* parent) {
  MultiGroup* g = parent->multi;
  if (!g) return;
  // the children are destroyed on their own threads (their HIP objects belong to those devices), then the workers stop
  if (!g->workers.empty()) {
    g->run([&](atmrt_ctx* k, int i) {
      if ((size_t)i < g->nccl.size() && g->nccl[(size_t)i]) (void)rccl()->CommDestroy(g->nccl[(size_t)i]);
      if ((size_t)i < g->nccl.size()) g->nccl[(size_t)i] = nullptr;
      if (k && k->comm) k->comm->nccl = nullptr;
      atmrt_ctx_destroy(k);
      g->kids[(size_t)i] = nullptr;
      return 0;
    });
    {
      std::lock_guard<std::mutex> lk(g->m);
      g->stop = true;
    }
    g->cv_task.notify_all();
    for (std::thread& t : g->workers) t.join();
  } else {
    for (atmrt_ctx* k : g->kids) atmrt_ctx_destroy(k);
  }
  delete g;
  parent->multi = nullptr;
}

// direct device-to-device copies where the topology allows them (the peer-copy route)
static void multi_enable_peer_access(MultiGroup* g) {
  const size_t n = g->devices.size();
  for (size_t i = 0; i < n; i++)
    for (size_t j = 0; j < n; j++) {
      int can = 0;
      if (g->devices[i] == g->devices[j] || hipDeviceCanAccessPeer(&can, g->devices[i], g->devices[j]) != hipSuccess || !can) continue;
      if (hipSetDevice(g->devices[i]) == hipSuccess) (void)hipDeviceEnablePeerAccess(g->devices[j], 0);
      (void)hipGetLastError(); // "already enabled" is fine
    }
  (void)hipSetDevice(g->devices[0]);
}

// RCCL let the context down (set-up, its probe collective, or an all-gather that returned an error): the devices of one process
// can always exchange their tiles by device-to-device copies.  The communicators stay where they are until the context goes.
static void multi_demote_to_peer(atmrt_ctx* parent) {
  MultiGroup* g = parent->multi;
  for (atmrt_ctx* k : g->kids) {
    k->comm->nccl = nullptr;
    k->comm->route = ATMRT_ROUTE_PEER;
  }
  g->demoted = parent->error;
  multi_enable_peer_access(g);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
extern "C" int atmrt_comm_unique_id(uint8_t id[ATMRT_COMM_ID_BYTES]) {
  if (!id) return ATMRT_ERR_INVALID_ARGUMENT;
  static_assert(sizeof(ncclUniqueId) == ATMRT_COMM_ID_BYTES, "the ABI hands an ncclUniqueId through as bytes");
  Rccl* r = rccl();
  if (!r->handle) return api_create_fail(ATMRT_ERR_NO_DEVICE, "RCCL is not available: " + r->error);
  ncclUniqueId u;
  ncclResult_t rc = r->GetUniqueId(&u);
  if (rc != ncclSuccess) return api_create_fail(ATMRT_ERR_HIP, std::string("ncclGetUniqueId: ") + r->GetErrorString(rc));
  memcpy(id, &u, sizeof u);
  return ATMRT_OK;
}

extern "C" int atmrt_comm_available(void) { return rccl()->handle ? 1 : 0; }

// re-tile after unbalanced frames?  ATMRT_TILE_BALANCE=0 / 1 overrides the default of the launch model
static bool balance_default(bool fallback) {
  const char* env = getenv("ATMRT_TILE_BALANCE");
  if (env && *env) return atoi(env) != 0;
  return fallback;
}

extern "C" int atmrt_ctx_comm_init_rank(atmrt_ctx* c, const uint8_t id[ATMRT_COMM_ID_BYTES], int32_t rank, int32_t world) {
  if (!c || !id) return ATMRT_ERR_INVALID_ARGUMENT;
  Rccl* r = rccl();
  if (!r->handle) return c->fail(ATMRT_ERR_NO_DEVICE, "RCCL is not available: %s", r->error.c_str());
  int rc = comm_attach(c, rank, world);
  if (rc) return rc;
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  hipError_t e = hipSetDevice(c->device);
  ncclResult_t nr = e == hipSuccess ? r->CommInitRank(&c->comm->nccl, world, u, rank) : ncclUnhandledCudaError;
  if (nr != ncclSuccess) {
    c->comm->nccl = nullptr;
    comm_destroy(c);
    return c->fail(ATMRT_ERR_HIP, "ncclCommInitRank(rank %d of %d) failed: %s", rank, world, r->GetErrorString(nr));
  }
  c->comm->route = ATMRT_ROUTE_RCCL;
  c->comm->balance = balance_default(world > 1);
  // the communicator's first collective runs HERE, where the host can still take another transport (atmrt_ctx_comm_init_external*)
  if ((rc = comm_probe(c))) {
    const std::string why = c->error;
    comm_destroy(c);
    return c->fail(rc, "%s", why.c_str());
  }
  return ATMRT_OK;
}

extern "C" int atmrt_ctx_comm_init_external(atmrt_ctx* c, int32_t rank, int32_t world, atmrt_all_gather_fn fn, void* user) {
  if (!c || !fn) return ATMRT_ERR_INVALID_ARGUMENT;
  int rc = comm_attach(c, rank, world);
  if (rc) return rc;
  c->comm->ext = fn;
  c->comm->ext_user = user;
  c->comm->route = ATMRT_ROUTE_EXTERNAL;
  c->comm->balance = balance_default(false); // ranks that may share a device (a rehearsal): their times say nothing
  return ATMRT_OK;
}

extern "C" int atmrt_ctx_comm_init_external_device(atmrt_ctx* c, int32_t rank, int32_t world, atmrt_all_gather_device_fn fn, void* user) {
  if (!c || !fn) return ATMRT_ERR_INVALID_ARGUMENT;
  int rc = comm_attach(c, rank, world);
  if (rc) return rc;
  c->comm->ext_dev = fn;
  c->comm->ext_user = user;
  c->comm->route = ATMRT_ROUTE_EXTERNAL_DEVICE;
  c->comm->balance = balance_default(world > 1);
  return ATMRT_OK;
}

extern "C" int atmrt_ctx_device_count(const atmrt_ctx* c) { return c ? multi_size(c) : 0; }

extern "C" int atmrt_ctx_create_multi(atmrt_ctx** out, const int32_t* devices, int32_t n) {
  if (!out) return api_create_fail(ATMRT_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  if (!devices || n < 1 || n > 64) return api_create_fail(ATMRT_ERR_INVALID_ARGUMENT, "a multi-device context needs 1..64 devices");
  atmrt_ctx* parent = nullptr;
  int rc = api_create_plain(&parent, devices[0]);
  if (rc) return rc;
  MultiGroup* g = new MultiGroup();
  parent->multi = g;
  g->devices.assign(devices, devices + n);
  g->kids.assign((size_t)n, nullptr);
  g->rc.assign((size_t)n, 0);
  g->recv_ptr.assign((size_t)n, nullptr);
  for (int i = 0; i < n; i++) {
    if ((rc = api_create_plain(&g->kids[(size_t)i], devices[i]))) { // leaves the message for atmrt_last_error(NULL)
      g->kids.resize((size_t)i);
      atmrt_ctx_destroy(parent);
      return rc;
    }
    atmrt_ctx* k = g->kids[(size_t)i];
    k->terrain = parent->terrain; // one tile store, one mosaic copy per device
    k->comm = new Comm();
    k->comm->rank = i;
    k->comm->world = n;
    k->comm->group = g;
    k->comm->route = n > 1 ? ATMRT_ROUTE_PEER : ATMRT_ROUTE_NONE;
  }
  // RCCL when the devices are distinct (it refuses two ranks on one device) and the library is there
  const char* want = getenv("ATMRT_GATHER");
  bool distinct = true;
  for (int i = 0; i < n; i++)
    for (int j = 0; j < i; j++) distinct = distinct && devices[i] != devices[j];
  const bool force_rccl = want && !strcmp(want, "rccl"), force_peer = want && !strcmp(want, "peer");
  if (!force_peer && (distinct || force_rccl)) {
    Rccl* r = rccl();
    std::string why;
    if (!r->handle) why = "RCCL is not available: " + r->error;
    else if (!distinct) why = "RCCL needs distinct devices";
    else {
      g->nccl.assign((size_t)n, nullptr);
      std::vector<int> devs(devices, devices + n);
      ncclResult_t nr = r->CommInitAll(g->nccl.data(), n, devs.data());
      if (nr != ncclSuccess) {
        why = std::string("ncclCommInitAll: ") + r->GetErrorString(nr);
        g->nccl.clear();
      } else {
        for (int i = 0; i < n; i++) {
          g->kids[(size_t)i]->comm->nccl = g->nccl[(size_t)i];
          g->kids[(size_t)i]->comm->route = ATMRT_ROUTE_RCCL;
        }
      }
    }
    if (!why.empty() && force_rccl) {
      atmrt_ctx_destroy(parent);
      return api_create_fail(ATMRT_ERR_HIP, "ATMRT_GATHER=rccl: " + why);
    }
  }
  for (int i = 0; i < n; i++) g->kids[(size_t)i]->comm->balance = balance_default(distinct && n > 1);
  for (int i = 0; i < n; i++) g->workers.emplace_back([g, i] { g->worker(i); });
  if (g->kids[0]->comm->route == ATMRT_ROUTE_RCCL && n > 1) {
    // the communicators' first collective, now: a failure here (or a hang: comm_probe's watchdog) falls back to peer copies
    rc = multi_forward(parent, [&](atmrt_ctx* k) { return comm_probe(k); });
    if (rc) {
      if (force_rccl) {
        const std::string why = parent->error;
        atmrt_ctx_destroy(parent);
        return api_create_fail(ATMRT_ERR_HIP, "ATMRT_GATHER=rccl: " + why);
      }
      multi_demote_to_peer(parent);
    }
  }
  if (n > 1 && g->kids[0]->comm->route == ATMRT_ROUTE_PEER) multi_enable_peer_access(g);
  *out = parent;
  return ATMRT_OK;
}

extern "C" int atmrt_generate_image_device(atmrt_ctx* c, const atmrt_device_planes_t* image, uint64_t* ray_steps, double* device_ms) {
  if (!c || !image) return ATMRT_ERR_INVALID_ARGUMENT;
  if (c->multi) {
    MultiGroup* g = c->multi;
    const size_t n = g->kids.size();
    for (size_t i = 0; i < n; i++)
      if (image[i].azimuth && !image_planes_complete(image[i]))
        return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "image %zu: every plane pointer must be a device allocation (or azimuth NULL to skip the device)", i);
    std::vector<uint64_t> steps(n, 0);
    std::vector<double> ms(n, 0.0);
    int rc = multi_forward(c, [&](atmrt_ctx* k) { return tile_generate(k, &steps[(size_t)k->comm->rank], &ms[(size_t)k->comm->rank]); });
    if (rc) return rc; // no collective has started: nobody waits for the rank that failed
    rc = multi_forward(c, [&](atmrt_ctx* k) { return tile_exchange(k, &image[(size_t)k->comm->rank]); });
    if (rc && n > 1 && g->kids[0]->comm->route == ATMRT_ROUTE_RCCL && !(getenv("ATMRT_GATHER") && !strcmp(getenv("ATMRT_GATHER"), "rccl"))) {
      // ncclAllGather returned an error (every device was released at the gate or came back with it): the tiles are still in
      // their slabs — exchange them by device-to-device copies, now and from now on
      multi_demote_to_peer(c);
      rc = multi_forward(c, [&](atmrt_ctx* k) { return tile_exchange(k, &image[(size_t)k->comm->rank]); });
    }
    if (rc) return rc;
    atmrt_comm_timings_t tm = g->kids[0]->comm->tm;
    uint64_t total = 0;
    for (size_t i = 0; i < n; i++) {
      const atmrt_comm_timings_t& t = g->kids[i]->comm->tm;
      tm.gather_ms = std::max(tm.gather_ms, t.gather_ms);
      tm.assemble_ms = std::max(tm.assemble_ms, t.assemble_ms);
      tm.tile_ms_max = std::max(tm.tile_ms_max, t.tile_ms_max);
      tm.tile_ms_min = std::min(tm.tile_ms_min, t.tile_ms_min);
      total += steps[i];
    }
    g->tm = tm;
    if (ray_steps) *ray_steps = total;
    if (device_ms) *device_ms = *std::max_element(ms.begin(), ms.end());
    return ATMRT_OK;
  }
  if (!image_planes_complete(*image)) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "every plane pointer must be a device allocation");
  if (!c->comm) return atmrt_generate_device(c, image, ray_steps, device_ms); // one device: the image is the tile
  int rc = tile_generate(c, ray_steps, device_ms);
  if (rc) return rc;
  return tile_exchange(c, image);
}

extern "C" int atmrt_image_hits_device(atmrt_ctx* c, const atmrt_device_hits_t* dst, uint64_t* n_hits) {
  if (!c) return ATMRT_ERR_INVALID_ARGUMENT;
  if (c->multi) {
    MultiGroup* g = c->multi;
    if (!dst) { // the total: every device holds it since the frame's own collective
      int rc = tile_hits(g->kids[0], nullptr, n_hits);
      if (rc) c->error = g->kids[0]->error;
      return rc;
    }
    // every device takes part in the lists' collective, whatever the caller wants on it (hit_offset NULL: nothing)
    std::vector<uint64_t> totals(g->kids.size(), 0);
    int rc = multi_forward(c, [&](atmrt_ctx* k) {
      const size_t i = (size_t)k->comm->rank;
      return tile_hits(k, &dst[i], &totals[i]);
    });
    if (n_hits) *n_hits = totals[0];
    g->tm.collectives = g->kids[0]->comm->tm.collectives;
    return rc;
  }
  if (!c->comm) return atmrt_last_hits_device(c, dst, n_hits); // one device: the tile's lists are the image's
  return tile_hits(c, dst, n_hits);
}

extern "C" int atmrt_draw_image_gathered_device(atmrt_ctx* c, const atmrt_coloring_t* coloring, uint8_t* const* rgb_device) {
  if (!c || !coloring || !rgb_device) return ATMRT_ERR_INVALID_ARGUMENT;
  if (c->multi) return multi_forward(c, [&](atmrt_ctx* k) { return tile_draw(k, coloring, rgb_device[(size_t)k->comm->rank]); });
  if (!c->comm) return rgb_device[0] ? atmrt_draw_image_device(c, coloring, rgb_device[0]) : ATMRT_ERR_INVALID_ARGUMENT;
  if (!c->last_valid) return c->fail(ATMRT_ERR_STATE, "atmrt_draw_image_gathered_device needs a frame: call atmrt_generate_image_device first");
  return tile_draw(c, coloring, rgb_device[0]);
}

extern "C" int atmrt_last_comm_timings(atmrt_ctx* c, atmrt_comm_timings_t* out) {
  if (!c || !out) return ATMRT_ERR_INVALID_ARGUMENT;
  if (c->multi) *out = c->multi->tm;
  else if (c->comm) *out = c->comm->tm;
  else {
    *out = atmrt_comm_timings_t{};
    out->world = 1;
    out->tile_ms_max = out->tile_ms_min = c->timings.total_ms;
  }
  return ATMRT_OK;
}

extern "C" int atmrt_ctx_tile_columns(atmrt_ctx* c, int32_t index, int32_t* col_begin, int32_t* col_end) {
  if (!c || !col_begin || !col_end) return ATMRT_ERR_INVALID_ARGUMENT;
  const Comm* cm = c->multi ? (index >= 0 && index < multi_size(c) ? multi_child(c, index)->comm : nullptr) : c->comm;
  if (c->multi && !cm) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "device index %d outside the context's %d devices", index, multi_size(c));
  if (!cm) { // a plain context: the whole width (or the caller's own shard)
    if (!c->have_params) return c->fail(ATMRT_ERR_STATE, "atmrt_set_params has not been called");
    const bool whole = c->params.col_begin == 0 && c->params.col_end == 0;
    *col_begin = whole ? 0 : c->params.col_begin;
    *col_end = whole ? c->params.width : c->params.col_end;
    return ATMRT_OK;
  }
  const int g = c->multi ? cm->rank : (index < 0 ? cm->rank : index);
  if (g < 0 || g >= cm->world) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "rank %d outside the %d ranks", g, cm->world);
  const std::vector<int>& cols = cm->exchanged ? cm->cols_frame : cm->cols;
  if ((int)cols.size() != cm->world + 1) return c->fail(ATMRT_ERR_STATE, "no frame has been tiled yet");
  *col_begin = cols[(size_t)g];
  *col_end = cols[(size_t)g + 1];
  return ATMRT_OK;
}

extern "C" int atmrt_tiles_rebalance(int32_t width, int32_t n_tiles, const int32_t* cols, const double* tile_ms, int32_t* cols_out) {
  if (!cols || !tile_ms || !cols_out || n_tiles < 1 || width < n_tiles) return ATMRT_ERR_INVALID_ARGUMENT;
  return tiles_rebalance(width, n_tiles, cols, tile_ms, cols_out) >= 0 ? ATMRT_OK : ATMRT_ERR_INVALID_ARGUMENT;
}

extern "C" int atmrt_debug_set_tiling(atmrt_ctx* c, const int32_t* cols, int32_t n) {
  if (!c) return ATMRT_ERR_INVALID_ARGUMENT;
  auto set = [&](atmrt_ctx* k) {
    Comm* cm = k->comm;
    if (!cols) { // back to the library's own tiling
      cm->cols_W = -1;
      cm->cols_pinned = false;
      return (int)ATMRT_OK;
    }
    if (n != cm->world + 1 || cols[0] != 0) return k->fail(ATMRT_ERR_INVALID_ARGUMENT, "a tiling of %d ranks is %d ascending boundaries from 0 to the width", cm->world, cm->world + 1);
    for (int g = 0; g < cm->world; g++)
      if (cols[g + 1] <= cols[g]) return k->fail(ATMRT_ERR_INVALID_ARGUMENT, "tile %d is empty", g);
    cm->cols.assign(cols, cols + n);
    cm->cols_W = cols[n - 1];
    cm->cols_pinned = true;
    return (int)ATMRT_OK;
  };
  if (c->multi) {
    for (int i = 0; i < multi_size(c); i++) {
      const int rc = set(multi_child(c, i));
      if (rc) {
        c->error = multi_child(c, i)->error;
        return rc;
      }
    }
    return ATMRT_OK;
  }
  if (!c->comm) return c->fail(ATMRT_ERR_STATE, "a plain context has no tiling: use col_begin / col_end");
  return set(c);
}

extern "C" int atmrt_debug_fail_next_collective(atmrt_ctx* c, int32_t index, int32_t nth) {
  if (!c || nth < 0) return ATMRT_ERR_INVALID_ARGUMENT;
  atmrt_ctx* k = c;
  if (c->multi) {
    if (index < 0 || index >= multi_size(c)) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "device index %d outside the context's %d devices", index, multi_size(c));
    k = multi_child(c, index);
  }
  if (!k->comm) return c->fail(ATMRT_ERR_STATE, "a plain context has no collective to fail");
  k->comm->fail_countdown = nth;
  return ATMRT_OK;
}

// ---------------------------------------------------------------------------------------------
// multi-device context, host consumer: Generator::generate -> Vec<Vec<ResultPixel>> in host memory
// ---------------------------------------------------------------------------------------------
int atmrt::multi_generate(atmrt_ctx* parent, atmrt_result_t* out) {
  MultiGroup* g = parent->multi;
  const size_t n = g->kids.size();
  if (!parent->have_params) return parent->fail(ATMRT_ERR_STATE, "atmrt_set_params has not been called");
  const int W = parent->params.width, H = parent->params.height;
  if (W < (int)n) return parent->fail(ATMRT_ERR_INVALID_ARGUMENT, "image width %d is less than the %zu devices", W, n);
  std::vector<uint64_t> nh(n, 0), steps(n, 0);
  std::vector<double> ms(n, 0.0);
  // (1) every device marches its tile; planes and packed lists stay in its HBM
  int rc = multi_forward(parent, [&](atmrt_ctx* k) {
    const size_t i = (size_t)k->comm->rank;
    int r = ensure_comm_events(k);
    if (r) return r;
    k->comm->image_valid = false;
    k->comm->exchanged = false;
    return api_generate_tile(k, nullptr, true, &nh[i], &steps[i], &ms[i]);
  });
  if (rc) return rc;
  uint64_t n_hits = 0;
  for (uint64_t v : nh) n_hits += v;
  if (atmrt_internal_result_alloc(out, (uint32_t)W, (uint32_t)H, n_hits))
    return parent->fail(ATMRT_ERR_INVALID_ARGUMENT, "out of host memory for %dx%d pixels / %llu hits", W, H, (unsigned long long)n_hits);
  // (2) planes: strided copies straight into the [H][W] block; lists: into this device's staging; row totals for the merge
  rc = multi_forward(parent, [&](atmrt_ctx* k) {
    Comm* cm = k->comm;
    const size_t i = (size_t)cm->rank;
    hipStream_t s = k->stream;
    HIP_TRY(k, hipSetDevice(k->device));
    const int c0 = k->last_c0, wl = k->last_wl;
    const size_t tile_px = (size_t)wl * H;
    const DensePlanes& d = k->last_dense;
    HIP_TRY(k, hipEventRecord(cm->ev_g0, s));
    HIP_TRY(k, hipMemcpy2DAsync(out->azimuth + c0, (size_t)W * 8, d.azimuth, (size_t)wl * 8, (size_t)wl * 8, (size_t)H, hipMemcpyDeviceToHost, s));
    HIP_TRY(k, hipMemcpy2DAsync(out->elevation_angle + c0, (size_t)W * 8, d.elevation_angle, (size_t)wl * 8, (size_t)wl * 8, (size_t)H, hipMemcpyDeviceToHost, s));
    HIP_TRY(k, hipMemcpy2DAsync(out->hit_count + c0, (size_t)W * 4, d.hit_count, (size_t)wl * 4, (size_t)wl * 4, (size_t)H, hipMemcpyDeviceToHost, s));
    const uint64_t m = nh[i];
    // staging: local offsets [H * wl] + the eight list arrays
    const size_t off_bytes = pad256(tile_px * 8);
    HIP_TRY(k, cm->h_stage.reserve(off_bytes + 5 * pad256(m * 8) + pad256(m * 24) + pad256(m * 32) + pad256(m * 4) + 256));
    char* st = cm->h_stage.as<char>();
    HIP_TRY(k, hipMemcpyAsync(st, k->last_offset, tile_px * 8, hipMemcpyDeviceToHost, s));
    char* p = st + off_bytes;
    const PackedHits& h = k->last_hits;
    const void* src[8] = {h.lat, h.lon, h.distance, h.elevation, h.path_length, h.normal, h.rgba, h.color_tag};
    const size_t width[8] = {8, 8, 8, 8, 8, 24, 32, 4};
    for (int a = 0; a < 8; a++) {
      if (m) HIP_TRY(k, hipMemcpyAsync(p, src[a], m * width[a], hipMemcpyDeviceToHost, s));
      p += pad256(m * width[a]);
    }
    HIP_TRY(k, hipEventRecord(cm->ev_g1, s));
    HIP_TRY(k, hipStreamSynchronize(s));
    const uint64_t* loc = reinterpret_cast<const uint64_t*>(st);
    cm->row_total.resize((size_t)H);
    for (int y = 0; y < H; y++) {
      const uint64_t end = y + 1 < H ? loc[(size_t)(y + 1) * wl] : m;
      cm->row_total[(size_t)y] = end - loc[(size_t)y * wl];
    }
    return (int)ATMRT_OK;
  });
  if (rc) {
    atmrt_result_free(out);
    return rc;
  }
  // (3) where every (row, device) segment of the lists starts in the image's pixel order
  std::vector<uint64_t> seg((size_t)H * n);
  uint64_t run = 0;
  for (int y = 0; y < H; y++)
    for (size_t i = 0; i < n; i++) {
      seg[(size_t)y * n + i] = run;
      run += g->kids[i]->comm->row_total[(size_t)y];
    }
  // (4) every device's host thread moves its own segments and writes its pixels' offsets
  const auto merge_t0 = std::chrono::steady_clock::now();
  rc = multi_forward(parent, [&](atmrt_ctx* k) {
    Comm* cm = k->comm;
    const size_t i = (size_t)cm->rank;
    const int c0 = k->last_c0, wl = k->last_wl;
    const uint64_t m = nh[i];
    const char* st = cm->h_stage.as<char>();
    const uint64_t* loc = reinterpret_cast<const uint64_t*>(st);
    const char* p = st + pad256((size_t)wl * H * 8);
    char* dstv[8] = {(char*)out->lat, (char*)out->lon, (char*)out->distance, (char*)out->elevation, (char*)out->path_length,
                     (char*)out->normal, (char*)out->rgba, (char*)out->color_tag};
    const size_t width[8] = {8, 8, 8, 8, 8, 24, 32, 4};
    const char* srcv[8];
    for (int a = 0; a < 8; a++) {
      srcv[a] = p;
      p += pad256(m * width[a]);
    }
    for (int y = 0; y < H; y++) {
      const uint64_t s0 = loc[(size_t)y * wl], cnt = cm->row_total[(size_t)y], d0 = seg[(size_t)y * n + i];
      if (cnt)
        for (int a = 0; a < 8; a++) memcpy(dstv[a] + d0 * width[a], srcv[a] + s0 * width[a], cnt * width[a]);
      uint64_t* o = out->hit_offset + (size_t)y * W + c0;
      for (int x = 0; x < wl; x++) o[x] = d0 + (loc[(size_t)y * wl + x] - s0);
    }
    return (int)ATMRT_OK;
  });
  if (rc) {
    atmrt_result_free(out);
    return rc;
  }
  out->ray_steps = 0;
  for (uint64_t v : steps) out->ray_steps += v;
  out->device_ms = *std::max_element(ms.begin(), ms.end());
  atmrt_comm_timings_t tm{};
  tm.world = (int32_t)n;
  tm.route = ATMRT_ROUTE_HOST;
  tm.tile_ms_min = 1e300;
  tm.assemble_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - merge_t0).count(); // the host merge of the lists
  for (size_t i = 0; i < n; i++) {
    atmrt_ctx* k = g->kids[i];
    float v = 0.f;
    (void)hipSetDevice(k->device);
    if (hipEventElapsedTime(&v, k->comm->ev_g0, k->comm->ev_g1) == hipSuccess) tm.gather_ms = std::max(tm.gather_ms, (double)v);
    tm.tile_ms_max = std::max(tm.tile_ms_max, k->timings.total_ms);
    tm.tile_ms_min = std::min(tm.tile_ms_min, k->timings.total_ms);
    tm.bytes_per_rank = std::max<uint64_t>(tm.bytes_per_rank, (uint64_t)k->last_npx * 20 + nh[i] * 100);
  }
  g->tm = tm;
  // the next frame's tiling from this frame's tile times (what tile_exchange does from the gathered trailers)
  Comm* c0m = g->kids[0]->comm;
  if (c0m->balance && !c0m->cols_pinned && n > 1 && tm.tile_ms_max * (double)n > 0.0) {
    std::vector<double> tms(n);
    double sum = 0.0;
    for (size_t i = 0; i < n; i++) sum += (tms[i] = g->kids[i]->timings.total_ms);
    std::vector<int> next(n + 1);
    if (tm.tile_ms_max * (double)n > 1.01 * sum && tiles_rebalance(W, (int)n, c0m->cols.data(), tms.data(), next.data()) == 1)
      for (atmrt_ctx* k : g->kids) k->comm->cols = next;
  }
  return ATMRT_OK;
}

int atmrt::multi_draw_image(atmrt_ctx* parent, const atmrt_coloring_t* coloring, uint8_t* rgb) {
  const int W = parent->params.width, H = parent->params.height;
  return multi_forward(parent, [&](atmrt_ctx* k) {
    if (!k->last_valid) return k->fail(ATMRT_ERR_STATE, "atmrt_draw_image needs a frame: call atmrt_generate first");
    HIP_TRY(k, hipSetDevice(k->device));
    HIP_TRY(k, k->d_io.reserve(3 * k->last_npx + 256));
    int rc = atmrt_draw_image_device(k, coloring, k->d_io.as<uint8_t>());
    if (rc) return rc;
    const size_t wl = (size_t)k->last_wl;
    HIP_TRY(k, hipMemcpy2D(rgb + 3 * (size_t)k->last_c0, 3 * (size_t)W, k->d_io.ptr, 3 * wl, 3 * wl, (size_t)H, hipMemcpyDeviceToHost));
    return (int)ATMRT_OK;
  });
}

int atmrt::multi_last_timings(atmrt_ctx* parent, atmrt_timings_t* out) {
  atmrt_timings_t t{};
  for (atmrt_ctx* k : parent->multi->kids) { // times: the slowest device; counters: all devices
    const atmrt_timings_t& a = k->timings;
    t.total_ms = std::max(t.total_ms, a.total_ms);
    t.profile_ms = std::max(t.profile_ms, a.profile_ms);
    t.paths_ms = std::max(t.paths_ms, a.paths_ms);
    t.intersect_ms = std::max(t.intersect_ms, a.intersect_ms);
    t.march_ms = std::max(t.march_ms, a.march_ms);
    t.finalize_ms = std::max(t.finalize_ms, a.finalize_ms);
    t.pack_ms = std::max(t.pack_ms, a.pack_ms);
    t.ray_steps += a.ray_steps;
    t.n_hits += a.n_hits;
  }
  *out = t;
  return ATMRT_OK;
}

int atmrt::multi_last_stats(atmrt_ctx* parent, atmrt_frame_stats_t* out) {
  atmrt_frame_stats_t t{};
  for (atmrt_ctx* k : parent->multi->kids) {
    const atmrt_frame_stats_t& a = k->stats;
    t.unlisted_rays += a.unlisted_rays;
    t.unlisted_columns += a.unlisted_columns;
    t.retraced_pixels += a.retraced_pixels;
    t.big_steps += a.big_steps;
    t.big_blend_pixels += a.big_blend_pixels;
    t.terrain_lookups += a.terrain_lookups;
    t.object_rays += a.object_rays;
    t.object_steps += a.object_steps;
  }
  *out = t;
  return ATMRT_OK;
}

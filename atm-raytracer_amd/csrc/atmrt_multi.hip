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

struct Comm {
  int rank = 0, world = 1;
  int route = ATMRT_ROUTE_NONE;
  ncclComm_t nccl = nullptr;
  MultiGroup* group = nullptr; // the devices of one process (peer copies, host threads)
  atmrt_all_gather_fn ext = nullptr;
  atmrt_all_gather_device_fn ext_dev = nullptr;
  void* ext_user = nullptr;
  hipEvent_t ev_g0 = nullptr, ev_g1 = nullptr, ev_a1 = nullptr;
  DevBuf d_slab, d_gathered, d_small, d_hits_send, d_hits_recv, d_loc_off, d_scan_tmp, d_rgb_tile, d_rgb_all;
  HostBuf h_send, h_recv, h_stage;
  // geometry of the last frame
  int W = 0, H = 0, wl_max = 0;
  size_t slab_bytes = 0;
  DensePlanes slab_planes{};
  DensePlanes last_image{}; // where the last frame's [H][W] planes were assembled (caller-owned)
  bool image_valid = false;
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

  bool aborted = false; // a device failed inside a task that has barriers: the others stop waiting (guarded by bm)

  // All devices meet here; false: one of them failed since the task began, and nobody waits any more.
  bool barrier() {
    std::unique_lock<std::mutex> lk(bm);
    if (aborted) return false;
    const uint64_t gen = bgen;
    if (++bcount == (int)kids.size()) {
      bcount = 0;
      bgen++;
      bcv.notify_all();
    } else {
      bcv.wait(lk, [&] { return bgen != gen || aborted; });
    }
    return !aborted;
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

void comm_columns(const atmrt_ctx* c, int width, int* c0, int* c1) {
  if (!c->comm) return;
  *c0 = shard_begin(width, c->comm->rank, c->comm->world);
  *c1 = shard_begin(width, c->comm->rank + 1, c->comm->world);
}

int multi_size(const atmrt_ctx* parent) { return parent->multi ? (int)parent->multi->kids.size() : 1; }
atmrt_ctx* multi_child(atmrt_ctx* parent, int i) { return parent->multi->kids[(size_t)i]; }

int multi_forward(atmrt_ctx* parent, const std::function<int(atmrt_ctx*)>& fn) {
  MultiGroup* g = parent->multi;
  const int rc = g->run([&](atmrt_ctx* k, int) { return fn(k); });
  if (rc)
    for (size_t i = 0; i < g->kids.size(); i++)
      if (g->rc[i]) {
        parent->error = "device " + std::to_string(g->devices[i]) + ": " + g->kids[i]->error;
        break;
      }
  return rc;
}

} // namespace atmrt

// ---------------------------------------------------------------------------------------------
// kernels: rank-major tiles -> the row-major image
// ---------------------------------------------------------------------------------------------
namespace {

__device__ __forceinline__ int tile_begin(int width, int g, int world) { return (int)((long long)g * width / world); }
__device__ __forceinline__ int tile_of_column(int x, int width, int world) {
  int g = (int)((long long)x * world / width); // tile_begin(g) <= x always; it may be one tile short
  while (g + 1 < world && tile_begin(width, g + 1, world) <= x) g++;
  return g;
}

// gathered: [G] slabs of `slab_bytes`; the slab of tile g is exactly what the generators write for a tile of H x wl_g pixels: 10 f64
// planes [H][wl_g] back to back (the planar normal is the last three) and the u32 hit_count plane.  One thread per image pixel,
// lanes = adjacent columns: reads and writes are coalesced inside a tile.
__global__ __launch_bounds__(256) void k_assemble_image(const char* __restrict__ gathered, size_t slab_bytes, int W, int H, int G,
                                                         DensePlanes image) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  const size_t npx = (size_t)W * H;
  if (p >= npx) return;
  const int y = (int)(p / (size_t)W), x = (int)(p % (size_t)W);
  const int g = tile_of_column(x, W, G);
  const int c0 = tile_begin(W, g, G), wl = tile_begin(W, g + 1, G) - c0;
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
                                                       uint8_t* __restrict__ rgb) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (size_t)W * H) return;
  const int y = (int)(p / (size_t)W), x = (int)(p % (size_t)W);
  const int g = tile_of_column(x, W, G);
  const int c0 = tile_begin(W, g, G), wl = tile_begin(W, g + 1, G) - c0;
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
                                                      const uint32_t* __restrict__ hit_count, const uint64_t* __restrict__ img_off,
                                                      PackedHits out) {
  const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (p >= (size_t)W * H) return;
  const uint32_t cnt = hit_count[p];
  if (!cnt) return;
  const int y = (int)(p / (size_t)W), x = (int)(p % (size_t)W);
  const int g = tile_of_column(x, W, G);
  const int c0 = tile_begin(W, g, G), wl = tile_begin(W, g + 1, G) - c0;
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

// Every rank's `bytes` at `send` -> all of them, rank-major, at `recv` (both in this rank's HBM), ordered on c->stream.
int comm_all_gather(atmrt_ctx* c, const void* send, void* recv, size_t bytes) {
  Comm* cm = c->comm;
  hipStream_t s = c->stream;
  if (!cm || (cm->world == 1 && cm->route != ATMRT_ROUTE_EXTERNAL && cm->route != ATMRT_ROUTE_EXTERNAL_DEVICE)) {
    if (cm && cm->nccl) { // RCCL at world size 1: the same call as with 8 ranks
      NCCL_TRY(c, rccl()->AllGather(send, recv, bytes, ncclUint8, cm->nccl, s));
      return ATMRT_OK;
    }
    HIP_TRY(c, hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, s));
    return ATMRT_OK;
  }
  switch (cm->route) {
    case ATMRT_ROUTE_RCCL:
      NCCL_TRY(c, rccl()->AllGather(send, recv, bytes, ncclUint8, cm->nccl, s));
      return ATMRT_OK;
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

// geometry of the tiles of the frame the context is configured for
int frame_geometry(atmrt_ctx* c) {
  Comm* cm = c->comm;
  if (!c->have_params) return c->fail(ATMRT_ERR_STATE, "atmrt_set_params has not been called");
  const int W = c->params.width, H = c->params.height, G = cm->world;
  if (W < G) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "image width %d is less than the %d ranks: a rank would be left without a column", W, G);
  int wl_max = 0;
  for (int g = 0; g < G; g++) wl_max = std::max(wl_max, shard_begin(W, g + 1, G) - shard_begin(W, g, G));
  cm->W = W, cm->H = H, cm->wl_max = wl_max;
  cm->slab_bytes = pad256((size_t)H * wl_max * (N_F64_PLANES * 8 + 4));
  return ATMRT_OK;
}

// Phase A of a shared frame: this rank's tile into its slab — the nine planes the generators write, back to back in ONE buffer
// (84 B per pixel), so that one collective moves them all.
int tile_generate(atmrt_ctx* c, uint64_t* ray_steps, double* device_ms) {
  int rc = frame_geometry(c);
  if (rc) return rc;
  if ((rc = ensure_comm_events(c))) return rc;
  Comm* cm = c->comm;
  cm->image_valid = false;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, cm->d_slab.reserve(cm->slab_bytes));
  int c0 = 0, c1 = 0;
  comm_columns(c, cm->W, &c0, &c1);
  const size_t ps = (size_t)cm->H * (size_t)(c1 - c0);
  double* base = cm->d_slab.as<double>();
  DensePlanes& d = cm->slab_planes;
  d.azimuth = base, d.elevation_angle = base + ps, d.lat = base + 2 * ps, d.lon = base + 3 * ps, d.distance = base + 4 * ps;
  d.elevation = base + 5 * ps, d.path_length = base + 6 * ps, d.normal = base + 7 * ps; // planar [3][H][wl]
  d.hit_count = reinterpret_cast<uint32_t*>(base + N_F64_PLANES * ps);
  uint64_t nh = 0;
  return api_generate_tile(c, &cm->slab_planes, false, &nh, ray_steps, device_ms);
}

// Phase B: the collective + the permutation into `image` (planes on this rank's device; NULL azimuth: take part, assemble nothing).
int tile_exchange(atmrt_ctx* c, const atmrt_device_planes_t* image) {
  Comm* cm = c->comm;
  hipStream_t s = c->stream;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, cm->d_gathered.reserve(cm->slab_bytes * (size_t)cm->world));
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
                       cm->slab_bytes, cm->W, cm->H, cm->world, img);
    cm->last_image = img;
    cm->image_valid = true;
  }
  HIP_TRY(c, hipEventRecord(cm->ev_a1, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  HIP_TRY(c, hipGetLastError());
  float g_ms = 0.f, a_ms = 0.f;
  HIP_TRY(c, hipEventElapsedTime(&g_ms, cm->ev_g0, cm->ev_g1));
  HIP_TRY(c, hipEventElapsedTime(&a_ms, cm->ev_g1, cm->ev_a1));
  cm->tm = atmrt_comm_timings_t{};
  cm->tm.gather_ms = g_ms;
  cm->tm.assemble_ms = a_ms;
  cm->tm.tile_ms_max = cm->tm.tile_ms_min = c->timings.total_ms;
  cm->tm.bytes_per_rank = cm->slab_bytes;
  cm->tm.world = cm->world;
  cm->tm.route = cm->nccl ? ATMRT_ROUTE_RCCL : cm->route;
  cm->tm.collectives = 1;
  return ATMRT_OK;
}

bool image_planes_complete(const atmrt_device_planes_t& p) {
  return p.azimuth && p.elevation_angle && p.hit_count && p.lat && p.lon && p.distance && p.elevation && p.path_length && p.normal;
}

// The lists of the last shared frame on this rank: totals -> (optionally) the lists in the image's pixel order.
int tile_hits(atmrt_ctx* c, const atmrt_device_hits_t* dst, uint64_t* n_total_out) {
  Comm* cm = c->comm;
  hipStream_t s = c->stream;
  if (!c->last_valid) return c->fail(ATMRT_ERR_STATE, "atmrt_image_hits_device needs a frame: call atmrt_generate_image_device first");
  if (!c->last_packed)
    return c->fail(ATMRT_ERR_STATE, "the last frame holds first-hit planes only (opaque scene): its trace points are the planes themselves");
  if (dst && !cm->image_valid) return c->fail(ATMRT_ERR_STATE, "the image planes of the last frame were not assembled on this device");
  HIP_TRY(c, hipSetDevice(c->device));
  const int G = cm->world;
  // (1) every rank's total: 8 bytes each
  HIP_TRY(c, cm->d_small.reserve(256 + 8 * (size_t)G + 64));
  uint64_t* d_mine = cm->d_small.as<uint64_t>();
  uint64_t* d_all = d_mine + 32;
  const uint64_t n_local = c->last_nhits;
  HIP_TRY(c, hipMemcpyAsync(d_mine, &n_local, 8, hipMemcpyHostToDevice, s));
  int rc = comm_all_gather(c, d_mine, d_all, 8);
  if (rc) return rc;
  std::vector<uint64_t> totals((size_t)G);
  HIP_TRY(c, hipMemcpyAsync(totals.data(), d_all, 8 * (size_t)G, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s)); // the one host synchronisation: buffer sizes must be known
  uint64_t n_total = 0, n_cap = 1;
  for (uint64_t v : totals) n_total += v, n_cap = std::max(n_cap, v);
  if (n_total_out) *n_total_out = n_total;
  cm->tm.collectives = 2;
  if (!dst) return ATMRT_OK;
  if (dst->capacity < n_total)
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "capacity %llu is less than the %llu trace points of the image",
                   (unsigned long long)dst->capacity, (unsigned long long)n_total);
  if (!dst->hit_offset || !dst->lat || !dst->lon || !dst->distance || !dst->elevation || !dst->path_length || !dst->normal ||
      !dst->color_tag || !dst->rgba)
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "every array pointer must be a device allocation");
  // (2) this rank's lists into a block laid out for the largest rank, (3) one all-gather of the blocks
  const size_t bb = HitBlock::bytes(n_cap);
  HIP_TRY(c, cm->d_hits_send.reserve(bb));
  HIP_TRY(c, cm->d_hits_recv.reserve(bb * (size_t)G));
  char* blk = cm->d_hits_send.as<char>();
  const PackedHits& h = c->last_hits;
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
  if ((rc = comm_all_gather(c, blk, cm->d_hits_recv.ptr, bb))) return rc;
  cm->tm.collectives = 3;
  // (4) offsets: every rank's own (a scan of its hit_count plane inside the gathered slabs) and the image's
  const size_t ps = (size_t)cm->H * cm->wl_max, npx = (size_t)cm->W * cm->H; // stride of the per-rank offset tables
  HIP_TRY(c, cm->d_loc_off.reserve(ps * (size_t)G * 8));
  HIP_TRY(c, cm->d_scan_tmp.reserve((std::max(npx, ps) / 2048 + 4) * 8 + 64));
  uint64_t* tmp = cm->d_scan_tmp.as<uint64_t>();
  unsigned long long* total = reinterpret_cast<unsigned long long*>(d_mine + 8); // scratch: the scans' grand totals are not needed
  for (int g = 0; g < G; g++) {
    const size_t n = (size_t)cm->H * (size_t)(shard_begin(cm->W, g + 1, G) - shard_begin(cm->W, g, G));
    const uint32_t* counts = reinterpret_cast<const uint32_t*>(cm->d_gathered.as<char>() + (size_t)g * cm->slab_bytes + N_F64_PLANES * n * 8);
    launch_scan_u32(counts, n, tmp, cm->d_loc_off.as<uint64_t>() + (size_t)g * ps, total, s);
  }
  launch_scan_u32(cm->last_image.hit_count, npx, tmp, dst->hit_offset, total, s);
  PackedHits out;
  out.lat = dst->lat, out.lon = dst->lon, out.distance = dst->distance, out.elevation = dst->elevation;
  out.path_length = dst->path_length, out.normal = dst->normal, out.color_tag = dst->color_tag, out.rgba = dst->rgba;
  hipLaunchKernelGGL(k_gather_hits, dim3(blocks_for(npx)), dim3(256), 0, s, cm->d_hits_recv.as<char>(), bb, (size_t)n_cap,
                     cm->d_loc_off.as<uint64_t>(), ps, cm->W, cm->H, G, (const uint32_t*)cm->last_image.hit_count,
                     (const uint64_t*)dst->hit_offset, out);
  HIP_TRY(c, hipStreamSynchronize(s));
  HIP_TRY(c, hipGetLastError());
  return ATMRT_OK;
}

// renderer::draw_image of this rank's tile + the 3 B/pixel exchange
int tile_draw(atmrt_ctx* c, const atmrt_coloring_t* coloring, uint8_t* rgb_image) {
  Comm* cm = c->comm;
  hipStream_t s = c->stream;
  int rc = frame_geometry(c);
  if (rc) return rc;
  HIP_TRY(c, hipSetDevice(c->device));
  const size_t tile_bytes = pad256(3 * (size_t)cm->H * cm->wl_max);
  HIP_TRY(c, cm->d_rgb_tile.reserve(tile_bytes));
  HIP_TRY(c, cm->d_rgb_all.reserve(tile_bytes * (size_t)cm->world));
  if ((rc = atmrt_draw_image_device(c, coloring, cm->d_rgb_tile.as<uint8_t>()))) return rc;
  if ((rc = comm_all_gather(c, cm->d_rgb_tile.ptr, cm->d_rgb_all.ptr, tile_bytes))) return rc;
  if (rgb_image)
    hipLaunchKernelGGL(k_assemble_rgb, dim3(blocks_for((size_t)cm->W * cm->H)), dim3(256), 0, s, cm->d_rgb_all.as<uint8_t>(), tile_bytes,
                       cm->W, cm->H, cm->world, rgb_image);
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
      if (k && k->comm && k->comm->nccl) (void)rccl()->CommDestroy(k->comm->nccl);
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
  return ATMRT_OK;
}

extern "C" int atmrt_ctx_comm_init_external(atmrt_ctx* c, int32_t rank, int32_t world, atmrt_all_gather_fn fn, void* user) {
  if (!c || !fn) return ATMRT_ERR_INVALID_ARGUMENT;
  int rc = comm_attach(c, rank, world);
  if (rc) return rc;
  c->comm->ext = fn;
  c->comm->ext_user = user;
  c->comm->route = ATMRT_ROUTE_EXTERNAL;
  return ATMRT_OK;
}

extern "C" int atmrt_ctx_comm_init_external_device(atmrt_ctx* c, int32_t rank, int32_t world, atmrt_all_gather_device_fn fn, void* user) {
  if (!c || !fn) return ATMRT_ERR_INVALID_ARGUMENT;
  int rc = comm_attach(c, rank, world);
  if (rc) return rc;
  c->comm->ext_dev = fn;
  c->comm->ext_user = user;
  c->comm->route = ATMRT_ROUTE_EXTERNAL_DEVICE;
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
  if (n > 1 && g->kids[0]->comm->route == ATMRT_ROUTE_PEER) { // direct device-to-device copies where the topology allows them
    for (int i = 0; i < n; i++)
      for (int j = 0; j < n; j++) {
        int can = 0;
        if (devices[i] == devices[j] || hipDeviceCanAccessPeer(&can, devices[i], devices[j]) != hipSuccess || !can) continue;
        if (hipSetDevice(devices[i]) == hipSuccess) (void)hipDeviceEnablePeerAccess(devices[j], 0);
        (void)hipGetLastError(); // "already enabled" is fine
      }
    (void)hipSetDevice(devices[0]);
  }
  for (int i = 0; i < n; i++) g->workers.emplace_back([g, i] { g->worker(i); });
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
    if ((rc = multi_forward(c, [&](atmrt_ctx* k) { return tile_exchange(k, &image[(size_t)k->comm->rank]); }))) return rc;
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
    std::vector<uint64_t> totals(c->multi->kids.size(), 0);
    int rc = multi_forward(c, [&](atmrt_ctx* k) {
      const size_t i = (size_t)k->comm->rank;
      return tile_hits(k, dst && dst[i].lat ? &dst[i] : nullptr, &totals[i]);
    });
    if (n_hits) *n_hits = totals[0];
    c->multi->tm.collectives = c->multi->kids[0]->comm->tm.collectives;
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
  }
  *out = t;
  return ATMRT_OK;
}

// atmrt_api.hip — implementation of the C ABI in include/atmrt.h: context, terrain store (own DTED
// parser), frame set-up and the launch sequence of each generator.  There is no CPU compute path:
// without a HIP device atmrt_ctx_create fails with ATMRT_ERR_NO_DEVICE.
#include <dirent.h>

#include <climits>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <utility>
#include <vector>

#include "atmrt_ctx.h"
#include "atmrt_hostmem.h"
#include "atmrt_kernels.h"
#include "atmrt_multi.h"
#include "atmrt_render.h"
#include "atmrt_tiff.h"

using namespace atmrt;

namespace {

std::mutex g_err_mutex;
std::string g_create_error = "";

} // namespace

// ---------------------------------------------------------------------------------------------
// DTED (MIL-PRF-89020B) — replaces crate dted 0.2's read_dted / read_dted_header (terrain/mod.rs:24,86)
// ---------------------------------------------------------------------------------------------
namespace {

constexpr long DTED_DATA_OFFSET = 3428; // UHL 80 + DSI 648 + ACC 2700

int parse_uint(const unsigned char* p, int n) {
  int v = 0;
  for (int i = 0; i < n; i++) {
    if (p[i] < '0' || p[i] > '9') return -1;
    v = v * 10 + (p[i] - '0');
  }
  return v;
}

bool parse_angle(const unsigned char* p, double* deg) { // DDDMMSSH
  int d = parse_uint(p, 3), m = parse_uint(p + 3, 2), s = parse_uint(p + 5, 2);
  if (d < 0 || m < 0 || s < 0) return false;
  double v = (double)d + (double)m / 60.0 + (double)s / 3600.0;
  if (p[7] == 'S' || p[7] == 'W') v = -v;
  else if (p[7] != 'N' && p[7] != 'E') return false;
  *deg = v;
  return true;
}

// returns 0, or a negative status with msg filled
int read_dted(const char* path, int* lat0, int* lon0, HostTile* tile, std::string* msg) {
  FILE* f = fopen(path, "rb");
  if (!f) {
    *msg = std::string("cannot open ") + path;
    return ATMRT_ERR_IO;
  }
  unsigned char uhl[80];
  double olat = 0, olon = 0;
  int rc = ATMRT_ERR_FORMAT;
  std::vector<unsigned char> rec;
  do {
    if (fread(uhl, 1, 80, f) != 80 || memcmp(uhl, "UHL1", 4) != 0) break;
    if (!parse_angle(uhl + 4, &olon) || !parse_angle(uhl + 12, &olat)) break;
    int nlon = parse_uint(uhl + 47, 4), nlat = parse_uint(uhl + 51, 4);
    if (nlon < 2 || nlat < 2) break;
    size_t rec_size = 12 + 2 * (size_t)nlat;
    rec.resize(rec_size);
    tile->n_lat = nlat;
    tile->n_lon = nlon;
    tile->posts.assign((size_t)nlat * nlon, 0);
    if (fseek(f, DTED_DATA_OFFSET, SEEK_SET)) break;
    bool ok = true;
    for (int j = 0; j < nlon && ok; j++) {
      if (fread(rec.data(), 1, rec_size, f) != rec_size || rec[0] != 0xAA) {
        ok = false;
        break;
      }
      for (int i = 0; i < nlat; i++) {
        unsigned v = ((unsigned)rec[8 + 2 * i] << 8) | rec[9 + 2 * i];
        int e = (int)(v & 0x7fff);
        if (v & 0x8000) e = -e; // signed magnitude
        tile->posts[(size_t)i * nlon + j] = (int16_t)e;
      }
    }
    if (!ok) break;
    // `f64::from(header.origin_lat) as i16`: truncation (terrain/mod.rs:91-92)
    *lat0 = sat_i16(olat);
    *lon0 = sat_i16(olon);
    rc = 0;
  } while (0);
  fclose(f);
  if (rc) *msg = std::string("Could not buffer terrain file ") + path; // terrain/mod.rs:117
  return rc;
}

} // namespace

// ---------------------------------------------------------------------------------------------
// lifetime
// ---------------------------------------------------------------------------------------------
extern "C" int atmrt_abi_version(void) { return ATMRT_ABI_VERSION; }

#ifndef ATMRT_SOURCE_HASH
#define ATMRT_SOURCE_HASH "unknown"
#endif
#ifndef ATMRT_BUILD_FLAGS
#define ATMRT_BUILD_FLAGS "unknown"
#endif
extern "C" const char* atmrt_build_info(void) { return "source_hash: " ATMRT_SOURCE_HASH "; " ATMRT_BUILD_FLAGS; }

extern "C" size_t atmrt_abi_sizeof(int which) {
  switch (which) {
    case 0: return sizeof(atmrt_params_t);
    case 1: return sizeof(atmrt_atmosphere_t);
    case 2: return sizeof(atmrt_object_t);
    case 3: return sizeof(atmrt_result_t);
    case 4: return sizeof(atmrt_device_planes_t);
    case 5: return sizeof(atmrt_earth_model_t);
    case 6: return sizeof(atmrt_position_t);
    case 7: return sizeof(atmrt_frame_t);
    case 8: return sizeof(atmrt_frame_stats_t);
    case 9: return sizeof(atmrt_timings_t);
    case 10: return sizeof(atmrt_coloring_t);
    case 11: return sizeof(atmrt_device_hits_t);
    case 12: return sizeof(atmrt_comm_timings_t);
    case 13: return sizeof(atmrt_temp_function_t);
    default: return 0;
  }
}

extern "C" const char* atmrt_last_error(const atmrt_ctx* ctx) {
  if (ctx) return ctx->error.c_str();
  std::lock_guard<std::mutex> lk(g_err_mutex);
  static thread_local std::string copy;
  copy = g_create_error;
  return copy.c_str();
}

static int create_fail(int code, const std::string& msg) {
  std::lock_guard<std::mutex> lk(g_err_mutex);
  g_create_error = msg;
  return code;
}

int atmrt::api_create_fail(int code, const std::string& msg) { return create_fail(code, msg); }

extern "C" int atmrt_ctx_create(atmrt_ctx** out, int device_ordinal) { return api_create_plain(out, device_ordinal); }

int atmrt::api_create_plain(atmrt_ctx** out, int device_ordinal) {
  if (!out) return create_fail(ATMRT_ERR_INVALID_ARGUMENT, "out is NULL");
  *out = nullptr;
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess || n <= 0)
    return create_fail(ATMRT_ERR_NO_DEVICE,
                       std::string("no HIP device available (") + hipGetErrorString(e) +
                           "); this library has no CPU path");
  if (device_ordinal < 0 || device_ordinal >= n)
    return create_fail(ATMRT_ERR_INVALID_ARGUMENT, "device ordinal out of range");
  if ((e = hipSetDevice(device_ordinal)) != hipSuccess)
    return create_fail(ATMRT_ERR_HIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  atmrt_ctx* c = new atmrt_ctx();
  c->device = device_ordinal;
  if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipStreamCreateWithFlags(&c->stream2, hipStreamNonBlocking)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming)) != hipSuccess ||
      (e = hipEventCreate(&c->ev_t0)) != hipSuccess || (e = hipEventCreate(&c->ev_t1)) != hipSuccess) {
    std::string msg = std::string("stream/event creation: ") + hipGetErrorString(e);
    atmrt_ctx_destroy(c);
    return create_fail(ATMRT_ERR_HIP, msg);
  }
  for (hipEvent_t& ev : c->ev_seg) {
    if ((e = hipEventCreateWithFlags(&ev, hipEventDisableTiming)) != hipSuccess) {
      std::string msg = std::string("hipEventCreate: ") + hipGetErrorString(e);
      atmrt_ctx_destroy(c);
      return create_fail(ATMRT_ERR_HIP, msg);
    }
  }
  for (hipEvent_t& ev : c->ev_scan) {
    if ((e = hipEventCreate(&ev)) != hipSuccess) {
      std::string msg = std::string("hipEventCreate: ") + hipGetErrorString(e);
      atmrt_ctx_destroy(c);
      return create_fail(ATMRT_ERR_HIP, msg);
    }
  }
  for (hipEvent_t& ev : c->ev) {
    if ((e = hipEventCreate(&ev)) != hipSuccess) {
      std::string msg = std::string("hipEventCreate: ") + hipGetErrorString(e);
      atmrt_ctx_destroy(c);
      return create_fail(ATMRT_ERR_HIP, msg);
    }
  }
  atmrt_params_default(&c->params);
  {
    atmrt_atmosphere_t us;
    atmrt_atmosphere_us76(&us);
    c->atm_def.assign(us);
  }
  *out = c;
  return ATMRT_OK;
}

extern "C" void atmrt_ctx_destroy(atmrt_ctx* c) {
  if (!c) return;
  if (c->multi) multi_destroy(c); // the children first, each on its own worker thread
  (void)hipSetDevice(c->device);
  if (c->comm) comm_destroy(c);
  if (c->stream) (void)hipStreamSynchronize(c->stream);
  if (c->stream2) (void)hipStreamSynchronize(c->stream2);
  for (DevBuf* b : {&c->d_posts, &c->d_tiles, &c->d_cells, &c->d_xs, &c->d_alt, &c->d_colcalc, &c->d_prof, &c->d_pelev,
                    &c->d_plen, &c->d_npath, &c->d_hit_step, &c->d_hit_offset, &c->d_scan_tmp, &c->d_counters,
                    &c->d_list_step, &c->d_list_pixel, &c->d_rect_rec, &c->d_objects, &c->d_textures, &c->d_plat, &c->d_plon,
                    &c->d_ccount, &c->d_coffset, &c->d_clist, &c->d_px_steps, &c->d_atm, &c->d_interp, &c->d_lat_dense, &c->d_lat_packed,
                    &c->d_lat_offset, &c->d_dense, &c->d_packed, &c->d_io, &c->d_slot_step, &c->d_slot_rec, &c->d_overflow, &c->d_slot_pixel, &c->d_slot_packed, &c->d_pelev_t, &c->d_plen_t, &c->d_col_cand, &c->d_col_ncand, &c->d_path_seg, &c->d_dprev, &c->d_step_prop, &c->d_blend_arena,
                    &c->d_object_rays, &c->d_slice, &c->d_overflow_arena})
    b->release(); // (a buffer missing from this list is released by its destructor when the context is deleted below)
  for (hipEvent_t ev : c->ev)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : c->ev_seg)
    if (ev) (void)hipEventDestroy(ev);
  for (hipEvent_t ev : c->ev_scan)
    if (ev) (void)hipEventDestroy(ev);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->ev_t0) (void)hipEventDestroy(c->ev_t0);
  if (c->ev_t1) (void)hipEventDestroy(c->ev_t1);
  if (c->stream) (void)hipStreamDestroy(c->stream);
  if (c->stream2) (void)hipStreamDestroy(c->stream2);
  delete c;
}

// ---------------------------------------------------------------------------------------------
// terrain
// ---------------------------------------------------------------------------------------------
extern "C" int atmrt_terrain_clear(atmrt_ctx* c) {
  if (!c) return ATMRT_ERR_INVALID_ARGUMENT;
  c->terrain->tiles.clear();
  c->terrain->generation++;
  return ATMRT_OK;
}

extern "C" int atmrt_terrain_add_tile(atmrt_ctx* c, int32_t lat0, int32_t lon0, int32_t n_lat, int32_t n_lon,
                                      const int16_t* posts) {
  if (!c) return ATMRT_ERR_INVALID_ARGUMENT;
  if (!posts || n_lat < 2 || n_lon < 2 || n_lat > 65536 || n_lon > 65536)
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "bad tile shape %d x %d", n_lat, n_lon);
  if (lat0 < -90 || lat0 > 89 || lon0 < -360 || lon0 > 359)
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "tile origin (%d, %d) out of range", lat0, lon0);
  HostTile& t = c->terrain->tiles[{lat0, lon0}]; // HashMap::insert replaces (terrain/mod.rs:93-96)
  t.n_lat = n_lat;
  t.n_lon = n_lon;
  t.posts.assign(posts, posts + (size_t)n_lat * n_lon);
  c->terrain->generation++;
  return ATMRT_OK;
}

// Terrain::from_folder, terrain/mod.rs:66-83
extern "C" int atmrt_terrain_load_dir(atmrt_ctx* c, const char* path, int32_t* n_files) {
  if (!c || !path) return ATMRT_ERR_INVALID_ARGUMENT;
  DIR* d = opendir(path);
  if (!d) return c->fail(ATMRT_ERR_IO, "Error opening the terrain data directory %s", path);
  int files = 0;
  while (struct dirent* ent = readdir(d)) {
    if (!strcmp(ent->d_name, ".") || !strcmp(ent->d_name, "..")) continue;
    std::string full = std::string(path) + "/" + ent->d_name;
    HostTile t;
    int lat0, lon0;
    std::string msg;
    int rc = read_dted(full.c_str(), &lat0, &lon0, &t, &msg);
    if (rc == ATMRT_ERR_FORMAT && atmrt_tiff::coords_from_name(ent->d_name, lat0, lon0)) {
      // not DTED, but named like a GeoTIFF tile (terrain/mod.rs:100-118 -> geotiff.rs).  The reference opens such a file lazily and
      // treats a failure as "no elevation here"; so does this loader: an undecodable file leaves its cell empty (0 m).
      std::string why;
      t = HostTile{};
      if (atmrt_tiff::read_dem(full, 3601, t.posts, why)) {
        t.n_lat = t.n_lon = 3601; // geotiff.rs:70-71: a 3600-interval grid per degree, file row = latitude index
        c->terrain->tiles[{lat0, lon0}] = std::move(t);
      } else {
        c->terrain->tiles.erase({lat0, lon0});
      }
      files++;
      continue;
    }
    if (rc) {
      closedir(d);
      c->terrain->generation++; // tiles read before the bad file are in the map: the mosaic must be rebuilt to match it
      return c->fail(rc, "%s", msg.c_str());
    }
    c->terrain->tiles[{lat0, lon0}] = std::move(t);
    files++;
  }
  closedir(d);
  c->terrain->generation++;
  if (n_files) *n_files = files;
  return ATMRT_OK;
}

// Upload the tile mosaic: all posts back to back + a dense (lat, lon) cell -> tile table.
static int upload_terrain(atmrt_ctx* c) {
  const TileStore& store = *c->terrain; // shared by the devices of a multi-device context: read-only while a frame is prepared
  if (c->terrain_uploaded == store.generation) return ATMRT_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  TerrainView tv{};
  if (store.tiles.empty()) {
    tv.lat_min = tv.lon_min = 0;
    tv.n_cells_lat = tv.n_cells_lon = 0;
    tv.skip_above = 1.0; // no tiles: every lookup is 0 m
    c->tv = tv;
    c->terrain_uploaded = store.generation;
    return ATMRT_OK;
  }
  int lat_min = 1 << 30, lat_max = -(1 << 30), lon_min = 1 << 30, lon_max = -(1 << 30);
  size_t total = 0;
  for (auto& kv : store.tiles) {
    lat_min = std::min(lat_min, kv.first.first);
    lat_max = std::max(lat_max, kv.first.first);
    lon_min = std::min(lon_min, kv.first.second);
    lon_max = std::max(lon_max, kv.first.second);
    total += kv.second.posts.size();
  }
  int ncl = lat_max - lat_min + 1, nco = lon_max - lon_min + 1;
  std::vector<int32_t> cells((size_t)ncl * nco, -1);
  std::vector<TileDesc> descs;
  std::vector<int16_t> mosaic;
  mosaic.reserve(total + 8);
  for (auto& kv : store.tiles) {
    TileDesc td;
    td.offset = (int64_t)mosaic.size();
    td.n_lat = kv.second.n_lat;
    td.n_lon = kv.second.n_lon;
    cells[(size_t)(kv.first.first - lat_min) * nco + (kv.first.second - lon_min)] = (int32_t)descs.size();
    descs.push_back(td);
    mosaic.insert(mosaic.end(), kv.second.posts.begin(), kv.second.posts.end());
  }
  HIP_TRY(c, c->d_posts.reserve(mosaic.size() * sizeof(int16_t)));
  HIP_TRY(c, c->d_tiles.reserve(descs.size() * sizeof(TileDesc)));
  HIP_TRY(c, c->d_cells.reserve(cells.size() * sizeof(int32_t)));
  HIP_TRY(c, hipMemcpy(c->d_posts.ptr, mosaic.data(), mosaic.size() * sizeof(int16_t), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_tiles.ptr, descs.data(), descs.size() * sizeof(TileDesc), hipMemcpyHostToDevice));
  HIP_TRY(c, hipMemcpy(c->d_cells.ptr, cells.data(), cells.size() * sizeof(int32_t), hipMemcpyHostToDevice));
  tv.posts = c->d_posts.as<int16_t>();
  tv.tiles = c->d_tiles.as<TileDesc>();
  tv.cell_tile = c->d_cells.as<int32_t>();
  tv.lat_min = lat_min;
  tv.lon_min = lon_min;
  tv.n_cells_lat = ncl;
  tv.n_cells_lon = nco;
  int16_t top = 0;
  for (int16_t v : mosaic) top = std::max(top, v);
  tv.skip_above = (double)top + 1.0;
  c->tv = tv;
  c->terrain_uploaded = store.generation;
  return ATMRT_OK;
}

// ---------------------------------------------------------------------------------------------
// configuration
// ---------------------------------------------------------------------------------------------
extern "C" void atmrt_params_default(atmrt_params_t* p) { // Config::default, params.rs:481-494
  memset(p, 0, sizeof *p);
  p->position.latitude = 0.0;
  p->position.longitude = 0.0;
  p->position.altitude_kind = ATMRT_ALT_RELATIVE; // params.rs:42-44
  p->position.altitude = 1.0;
  p->frame.direction = 0.0;
  p->frame.tilt = 0.0;
  p->frame.fov = 30.0;              // params.rs:156-158
  p->frame.max_distance = 150000.0; // params.rs:160-162
  p->earth.kind = ATMRT_EARTH_SPHERICAL;
  p->earth.radius = 6371000.0; // params.rs:467-471
  p->wavelength = 530e-9;      // params.rs:477-479
  p->simulation_step = 50.0;   // params.rs:473-475
  p->terrain_alpha = 1.0;      // params.rs:76-78
  p->straight_rays = 0;
  p->generator = ATMRT_GEN_FAST; // params.rs:427-429
  p->width = 640;                // params.rs:419-425
  p->height = 480;
}

extern "C" void atmrt_atmosphere_us76(atmrt_atmosphere_t* a) {
  static const double alt[7] = {0.0, 11000.0, 20000.0, 32000.0, 47000.0, 51000.0, 71000.0};
  static const double lapse[7] = {-0.0065, 0.0, 0.001, 0.0028, 0.0, -0.0028, -0.002};
  static const struct Table {
    atmrt_temp_function_t fn[7];
    Table() {
      memset(fn, 0, sizeof fn);
      for (int k = 0; k < 7; k++) {
        fn[k].kind = ATMRT_TEMP_LINEAR;
        fn[k].altitude = alt[k];
        fn[k].gradient = lapse[k];
      }
    }
  } table;
  memset(a, 0, sizeof *a);
  a->pressure_altitude = 0.0;
  a->pressure = 101325.0;
  a->temperature_altitude = 0.0;
  a->temperature = 288.15;
  a->has_temperature_fixed_point = 1;
  a->n_functions = 7;
  a->functions = table.fn; // library-owned, immutable
}

// everything an atmosphere definition says, as one byte string (the struct's scalars, its functions, their spline points)
static std::vector<uint8_t> atm_def_image(const atmrt_atmosphere_t& a) {
  std::vector<uint8_t> out;
  auto put = [&](const void* p, size_t n) { out.insert(out.end(), (const uint8_t*)p, (const uint8_t*)p + n); };
  atmrt_atmosphere_t head = a;
  head.functions = nullptr;
  put(&head, sizeof head);
  for (int j = 0; j < a.n_functions && a.functions; j++) {
    atmrt_temp_function_t fn = a.functions[j];
    const double *xs = fn.point_altitude, *ys = fn.point_temperature;
    fn.point_altitude = fn.point_temperature = nullptr;
    put(&fn, sizeof fn);
    if (fn.kind == ATMRT_TEMP_SPLINE && fn.n_points > 0 && xs && ys) {
      put(xs, sizeof(double) * (size_t)fn.n_points);
      put(ys, sizeof(double) * (size_t)fn.n_points);
    }
  }
  return out;
}

extern "C" int atmrt_set_atmosphere(atmrt_ctx* c, const atmrt_atmosphere_t* a) {
  if (!c || !a) return ATMRT_ERR_INVALID_ARGUMENT;
  AtmTableBuf t;
  static const char* why[] = {"", "bad function count or kind", "function altitudes must increase",
                              "no temperature anchor: give temperature_fixed_point or a Spline", "bad spline points"};
  int rc = atm_compile(*a, c->params.wavelength, t);
  if (rc) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "invalid atmosphere definition: %s", why[-rc <= 4 ? -rc : 1]);
  if (!(a->pressure > 0.0)) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "the pressure fixed point must be positive");
  // Nothing derived is validated: a profile that runs through 0 K, or whose hydrostatic pressure overflows, is marched like any
  // other (NaN and inf propagate as they do in the reference's f64 arithmetic); such segments get no certificate (atm_certify)
  // and are evaluated with IEEE operations.  Until round 2 they were rejected here.
  if (!c->atm_def_bytes.empty() && c->atm_def_bytes == atm_def_image(*a)) {
    // the same definition again (hosts set it before every frame): the compiled table of the last frame stands
  } else {
    c->atm_def.assign(*a);
    c->atm_def_bytes = atm_def_image(*a);
    c->atm_def_serial++;
  }
  if (c->multi) return multi_forward(c, [a](atmrt_ctx* k) { return atmrt_set_atmosphere(k, a); });
  return ATMRT_OK;
}

extern "C" int atmrt_set_params(atmrt_ctx* c, const atmrt_params_t* p) {
  if (!c || !p) return ATMRT_ERR_INVALID_ARGUMENT;
  Earth e;
  if (earth_resolve(p->earth, e)) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "unknown earth model kind %d", p->earth.kind);
  if ((e.calc == 2 && !(e.calc_radius > 0.0)) || (e.calc == 3 && !(e.a > 0.0 && e.b > 0.0)))
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "earth model radius / axes must be positive");
  if (!(p->simulation_step > 0.0) || !std::isfinite(p->simulation_step))
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "simulation_step must be positive");
  if (!(p->frame.max_distance > 0.0) || !std::isfinite(p->frame.max_distance))
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "max_distance must be positive and finite");
  if (p->frame.max_distance / p->simulation_step > 4.0e6)
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "max_distance / simulation_step exceeds 4e6 samples per ray");
  if (p->width == 0 || p->height == 0 || p->width > 32767 || p->height > 32767) // i16 casts, fast.rs:116,122
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "width and height must be in 1..32767");
  if (!(p->col_begin == 0 && p->col_end == 0) && !(p->col_begin < p->col_end && p->col_end <= p->width))
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "column shard [%u, %u) outside width %u", p->col_begin, p->col_end, p->width);
  if (p->generator < 0 || p->generator > 2) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "unknown generator %d", p->generator);
  if (!(p->wavelength > 0.0)) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "wavelength must be positive");
  if (p->simulation_step != c->params.simulation_step || p->frame.max_distance != c->params.frame.max_distance ||
      !c->have_params)
    c->xs_dirty = true;
  // distances handed to SphericalCalc are sums of steps and interpolation points inside a step: 0 or >= ~1e-17 step
  if (e.calc_radius >= 1.0e-30 && e.calc_radius <= 1.0e30 && p->simulation_step >= 1.0e-10 && p->frame.max_distance <= 1.0e30)
    e.flat_dirs |= EARTH_FAST_DIV;
  if ((c->comm || c->multi) && !(p->col_begin == 0 && p->col_end == 0))
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "col_begin / col_end must be 0 on a context that shares its frame with other ranks or "
                                               "devices: the library assigns the pixel-column tiles itself");
  c->params = *p;
  c->earth = e;
  c->have_params = true;
  if (c->multi) return multi_forward(c, [p](atmrt_ctx* k) { return atmrt_set_params(k, p); });
  return ATMRT_OK;
}

extern "C" int atmrt_objects_set(atmrt_ctx* c, const atmrt_object_t* objects, size_t n) {
  if (!c || (n && !objects)) return ATMRT_ERR_INVALID_ARGUMENT;
  if (n > 1000000) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "too many objects");
  std::vector<ObjectDev> objs(n);
  std::vector<uint8_t> pool;
  for (size_t i = 0; i < n; i++) {
    const atmrt_object_t& s = objects[i];
    ObjectDev& o = objs[i];
    memset(&o, 0, sizeof o);
    if (s.kind != ATMRT_OBJ_FRUSTUM && s.kind != ATMRT_OBJ_BILLBOARD)
      return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "object %zu: unknown kind %d", i, s.kind);
    if (s.position.altitude_kind != ATMRT_ALT_ABSOLUTE && s.position.altitude_kind != ATMRT_ALT_RELATIVE)
      return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "object %zu: unknown altitude kind", i);
    o.kind = s.kind;
    o._pad = s.position.altitude_kind;
    o.lat = s.position.latitude;
    o.lon = s.position.longitude;
    o.elev = s.position.altitude;
    o.r1 = s.r1;
    o.r2 = s.r2;
    o.height = s.height;
    o.width = s.width;
    for (int k = 0; k < 4; k++) o.color[k] = s.color[k];
    if (s.kind == ATMRT_OBJ_BILLBOARD) {
      // Image::get_pixel clamps to (0, w - 2): f64::clamp panics for textures smaller than 2x2 (object/mod.rs:95,100)
      if (!s.texture_rgba || s.texture_width < 2 || s.texture_height < 2 || s.texture_width > 16384 || s.texture_height > 16384)
        return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "object %zu: a billboard needs an RGBA8 texture of at least 2x2", i);
      o.tex_w = (int32_t)s.texture_width;
      o.tex_h = (int32_t)s.texture_height;
      o.tex_offset = (int64_t)pool.size();
      size_t bytes = (size_t)s.texture_width * s.texture_height * 4;
      pool.insert(pool.end(), s.texture_rgba, s.texture_rgba + bytes);
    }
  }
  c->objects.swap(objs);
  c->textures.swap(pool);
  c->objects_dirty = true;
  if (c->multi) return multi_forward(c, [objects, n](atmrt_ctx* k) { return atmrt_objects_set(k, objects, n); });
  return ATMRT_OK;
}

// ---------------------------------------------------------------------------------------------
// frame set-up
// ---------------------------------------------------------------------------------------------
static int prepare_frame(atmrt_ctx* c, Frame* out) {
  if (!c->have_params) return c->fail(ATMRT_ERR_STATE, "atmrt_set_params has not been called");
  HIP_TRY(c, hipSetDevice(c->device));
  int rc = upload_terrain(c);
  if (rc) return rc;
  const atmrt_params_t& p = c->params;
  // The compiled table and its certificate depend on the definition, the wavelength, the ODE's shape and the step: hosts set the
  // same atmosphere before every frame (the Python mirror does), and the certificate's bisections are a quarter of a millisecond
  // of host time — a twentieth of a Fast frame.  Recompiled, re-certified and uploaded again only when one of them changed.
  const atmrt_ctx::AtmKey key{c->atm_def_serial, p.wavelength, p.simulation_step, c->earth.shape_radius, c->earth.spherical, 0};
  const bool atm_fresh = !c->atm_key_valid || memcmp(&key, &c->atm_key, sizeof key) != 0;
  if (atm_fresh) {
    if (atm_compile(c->atm_def.pod, p.wavelength, c->atm)) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "invalid atmosphere");
    atm_certify(c->atm.table(), c->earth.spherical != 0, c->earth.shape_radius, p.simulation_step);
    if (getenv("ATMRT_NO_TIGHT")) // experiments: the voting path of dm_div3 on every segment (same bits, tests/test_gpu_march_variants.py)
      for (int k = 0; k < c->atm.table().n; k++) {
        c->atm.table().seg(k).flags &= ~ATM_SEG_TIGHT;
        c->atm.table().seg(k).tight_lo = INFINITY, c->atm.table().seg(k).tight_hi = -INFINITY;
      }
  }
  pinhole_init(p, c->pinhole);
  if (c->xs_dirty) {
    // distance table by repeated addition, exactly like `distance += step` (utils.rs:191-196) and the
    // stepper's x; n_t = #{k: xs[k] < max}; the path cache gets one element more than the first k
    // whose PREVIOUS x exceeds max (utils.rs:160-170)
    c->xs.clear();
    double d = 0.0;
    int n_t = 0;
    for (;;) {
      c->xs.push_back(d);
      if (d < p.frame.max_distance) n_t++;
      size_t k = c->xs.size() - 1; // index of d
      if (k >= 1 && c->xs[k - 1] > p.frame.max_distance) break;
      d += p.simulation_step;
      if (c->xs.size() > 5000000) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "distance table too long");
    }
    c->n_t = n_t;
    c->n_path_cap = (int)c->xs.size();
    HIP_TRY(c, c->d_xs.reserve(c->xs.size() * sizeof(double)));
    HIP_TRY(c, hipMemcpy(c->d_xs.ptr, c->xs.data(), c->xs.size() * sizeof(double), hipMemcpyHostToDevice));
    c->xs_dirty = false;
  }
  Frame f{};
  f.p = p;
  f.earth = c->earth;
  f.inv_shape_radius = c->earth.spherical && c->earth.shape_radius != 0.0 ? 1.0 / c->earth.shape_radius : 0.0;
  if (atm_fresh) {
    HIP_TRY(c, c->d_atm.reserve(c->atm.bytes()));
    HIP_TRY(c, hipMemcpy(c->d_atm.ptr, &c->atm.table(), c->atm.bytes(), hipMemcpyHostToDevice));
    c->atm_key = key;
    c->atm_key_valid = true;
  }
  f.atm = c->d_atm.as<AtmTable>();
  f.ph = c->pinhole;
  f.tv = c->tv;
  HIP_TRY(c, c->d_alt.reserve(sizeof(double)));
  f.alt = c->d_alt.as<double>();
  f.xs = c->d_xs.as<double>();
  // the device table is rewritten by k_resolve every frame (Altitude::abs depends on the terrain), so upload it each time
  if (!c->objects.empty()) {
    HIP_TRY(c, c->d_objects.reserve(c->objects.size() * sizeof(ObjectDev)));
    HIP_TRY(c, hipMemcpy(c->d_objects.ptr, c->objects.data(), c->objects.size() * sizeof(ObjectDev), hipMemcpyHostToDevice));
    if (c->objects_dirty && !c->textures.empty()) {
      HIP_TRY(c, c->d_textures.reserve(c->textures.size()));
      HIP_TRY(c, hipMemcpy(c->d_textures.ptr, c->textures.data(), c->textures.size(), hipMemcpyHostToDevice));
    }
    c->objects_dirty = false;
  }
  f.objects = c->objects.empty() ? nullptr : c->d_objects.as<ObjectDev>();
  f.textures = c->d_textures.as<uint8_t>();
  f.n_objects = (int32_t)c->objects.size();
  f.n_t = c->n_t;
  f.n_path_cap = c->n_path_cap;
  {
    int c0 = (p.col_begin == 0 && p.col_end == 0) ? 0 : p.col_begin;
    int c1 = (p.col_begin == 0 && p.col_end == 0) ? p.width : p.col_end;
    if (c->comm) comm_columns(c, p.width, &c0, &c1); // a rank / device of a shared frame: the library's own pixel-column tile
    if (c1 <= c0) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "image width %u leaves this rank without a pixel column", p.width);
    f.c0 = c0;
    f.wl = c1 - c0;
  }
  f.h = p.height;
  f.opaque = (p.terrain_alpha == 1.0 && c->objects.empty()) ? 1 : 0;
  f.lattice = 0;
  f.atm_cubic = atm_has_cubic(c->atm.table()) ? 1 : 0;
  f.di0 = f.ei0 = 0;
  f.dir_step = f.elev_step = 0.0;
  *out = f;
  return ATMRT_OK;
}

static PackedHits carve_packed(void* base, size_t n);
static size_t packed_bytes(size_t n);

static int prepare_workspace(atmrt_ctx* c, const Frame& f, Workspace* ws) {
  size_t npx = (size_t)f.wl * f.h;
  HIP_TRY(c, c->d_counters.reserve(N_COUNTERS * sizeof(uint64_t)));
  HIP_TRY(c, c->d_hit_step.reserve(npx * sizeof(int32_t)));
  HIP_TRY(c, c->d_hit_offset.reserve(npx * sizeof(uint64_t)));
  size_t nsamples = f.n_objects && f.p.generator != ATMRT_GEN_RECTILINEAR ? (size_t)f.n_t * f.wl : 0;
  HIP_TRY(c, c->d_scan_tmp.reserve((std::max(npx, nsamples) / 2048 + 2) * sizeof(uint64_t)));
  if (nsamples) {
    HIP_TRY(c, c->d_plat.reserve(nsamples * sizeof(double)));
    HIP_TRY(c, c->d_plon.reserve(nsamples * sizeof(double)));
    HIP_TRY(c, c->d_ccount.reserve(nsamples * sizeof(uint32_t)));
    HIP_TRY(c, c->d_coffset.reserve(nsamples * sizeof(uint64_t)));
  }
  ws->plat = c->d_plat.as<double>();
  ws->plon = c->d_plon.as<double>();
  ws->ccount = c->d_ccount.as<uint32_t>();
  ws->coffset = c->d_coffset.as<uint64_t>();
  ws->clist = c->d_clist.as<uint32_t>();
  ws->px_steps = nullptr;
  if (f.p.generator != ATMRT_GEN_RECTILINEAR) {
    HIP_TRY(c, c->d_colcalc.reserve((size_t)f.wl * sizeof(DirCalc)));
    HIP_TRY(c, c->d_prof.reserve((size_t)f.n_t * f.wl * sizeof(double)));
    HIP_TRY(c, c->d_pelev.reserve((size_t)f.h * f.n_path_cap * sizeof(double)));
    HIP_TRY(c, c->d_plen.reserve((size_t)f.h * f.n_path_cap * sizeof(double)));
    HIP_TRY(c, c->d_npath.reserve((size_t)f.h * sizeof(int32_t)));
    HIP_TRY(c, c->d_path_seg.reserve((size_t)f.h * sizeof(PathSegState)));
    if (f.n_objects == 0) HIP_TRY(c, c->d_dprev.reserve(npx * sizeof(double)));
  }
  if (f.p.generator == ATMRT_GEN_RECTILINEAR) HIP_TRY(c, c->d_rect_rec.reserve(4 * npx * sizeof(double)));
  ws->rect_rec = c->d_rect_rec.as<double>();
  ws->slot_packed = PackedHits{};
  if (!f.opaque || f.n_objects > 0) { // slots of the counting march / scan / trace passes
    HIP_TRY(c, c->d_slot_step.reserve((size_t)RECT_SLOTS * npx * sizeof(uint32_t)));
    if (f.p.generator == ATMRT_GEN_RECTILINEAR) HIP_TRY(c, c->d_slot_rec.reserve(4 * (size_t)RECT_SLOTS * npx * sizeof(double)));
    if (f.n_objects > 0) {
      HIP_TRY(c, c->d_slot_pixel.reserve((size_t)RECT_SLOTS * npx * sizeof(uint32_t)));
      HIP_TRY(c, c->d_slot_packed.reserve(packed_bytes((size_t)RECT_SLOTS * npx)));
      ws->slot_packed = carve_packed(c->d_slot_packed.ptr, (size_t)RECT_SLOTS * npx);
    }
  }
  ws->slot_pixel = c->d_slot_pixel.as<uint32_t>();
  ws->slot_step = c->d_slot_step.as<uint32_t>();
  ws->slot_rec = c->d_slot_rec.as<double>();
  ws->overflow = c->d_overflow.as<uint32_t>();
  ws->n_overflow = 0;
  ws->alt = c->d_alt.as<double>();
  ws->colcalc = c->d_colcalc.as<DirCalc>();
  ws->prof = c->d_prof.as<double>();
  ws->pelev = c->d_pelev.as<double>();
  ws->plen = c->d_plen.as<double>();
  if (f.p.generator != ATMRT_GEN_RECTILINEAR && f.n_objects > 0) {
    HIP_TRY(c, c->d_pelev_t.reserve((size_t)f.h * f.n_path_cap * sizeof(double)));
    HIP_TRY(c, c->d_plen_t.reserve((size_t)f.h * f.n_path_cap * sizeof(double)));
  }
  if (f.p.generator != ATMRT_GEN_RECTILINEAR && f.n_objects > 0) {
    HIP_TRY(c, c->d_col_cand.reserve((size_t)f.wl * 64 * sizeof(int32_t)));
    HIP_TRY(c, c->d_col_ncand.reserve((size_t)f.wl * sizeof(int32_t)));
    HIP_TRY(c, c->d_col_lo.reserve((size_t)f.wl * 64 * sizeof(double)));
    HIP_TRY(c, c->d_col_hi.reserve((size_t)f.wl * 64 * sizeof(double)));
    HIP_TRY(c, c->d_traced.reserve(npx));
  }
  ws->col_cand = c->d_col_cand.as<int32_t>();
  ws->col_ncand = c->d_col_ncand.as<int32_t>();
  ws->col_lo = c->d_col_lo.as<double>();
  ws->col_hi = c->d_col_hi.as<double>();
  ws->traced = c->d_traced.as<uint8_t>();
  ws->pelev_t = c->d_pelev_t.as<double>();
  ws->plen_t = c->d_plen_t.as<double>();
  ws->npath = c->d_npath.as<int32_t>();
  ws->path_seg = c->d_path_seg.as<PathSegState>();
  ws->dprev = c->d_dprev.as<double>();
  ws->hit_step = c->d_hit_step.as<int32_t>();
  ws->hit_offset = c->d_hit_offset.as<uint64_t>();
  ws->scan_tmp = c->d_scan_tmp.as<uint64_t>();
  ws->counters = c->d_counters.as<uint64_t>();
  ws->list_step = nullptr;
  ws->list_pixel = nullptr;
  ws->step_prop = nullptr;
  ws->object_rays = nullptr;
  ws->step_ctx = nullptr;
  ws->overflow_arena = nullptr;
  ws->overflow_cap = 0;
  ws->n_overflow_records = 0;
  ws->overflow_packed = PackedHits{};
  if (f.p.generator == ATMRT_GEN_RECTILINEAR && !f.opaque) { // the counting passes' trace points beyond the slots
    static const long forced_cap = [] { // test hook: a tiny arena forces the second-pass route (tests/test_gpu_march_variants.py)
      const char* e = getenv("ATMRT_OVERFLOW_CAP");
      return e ? atol(e) : -1L;
    }();
    ws->overflow_cap = forced_cap >= 0 ? (size_t)forced_cap : std::max<size_t>(65536, npx / 4);
    const size_t rec_bytes = (overflow_arena_bytes(ws->overflow_cap) + 255) / 256 * 256;
    HIP_TRY(c, c->d_overflow_arena.reserve(rec_bytes + (f.n_objects ? packed_bytes(ws->overflow_cap) : 0)));
    ws->overflow_arena = c->d_overflow_arena.as<char>();
    if (f.n_objects) ws->overflow_packed = carve_packed(ws->overflow_arena + rec_bytes, ws->overflow_cap);
  }
  ws->slice_state = nullptr;
  SliceLayout slices;
  if (march_slice_layout(f, slices)) { // a small Rectilinear launch without scene objects: the time-sliced march
    HIP_TRY(c, c->d_slice.reserve(slices.bytes));
    ws->slice_state = c->d_slice.as<char>();
  }
  return ATMRT_OK;
}

static DensePlanes carve_dense(void* base, size_t npx) {
  DensePlanes d;
  char* p = static_cast<char*>(base);
  auto take = [&](size_t bytes) {
    void* r = p;
    p += (bytes + 255) / 256 * 256;
    return r;
  };
  d.azimuth = (double*)take(npx * 8);
  d.elevation_angle = (double*)take(npx * 8);
  d.lat = (double*)take(npx * 8);
  d.lon = (double*)take(npx * 8);
  d.distance = (double*)take(npx * 8);
  d.elevation = (double*)take(npx * 8);
  d.path_length = (double*)take(npx * 8);
  d.normal = (double*)take(npx * 24);
  d.hit_count = (uint32_t*)take(npx * 4);
  return d;
}
static size_t dense_bytes(size_t npx) { return 10 * (npx * 8 + 256) + npx * 4 + 512; }

static PackedHits carve_packed(void* base, size_t n) {
  PackedHits h;
  char* p = static_cast<char*>(base);
  auto take = [&](size_t bytes) {
    void* r = p;
    p += (bytes + 255) / 256 * 256;
    return r;
  };
  h.lat = (double*)take(n * 8);
  h.lon = (double*)take(n * 8);
  h.distance = (double*)take(n * 8);
  h.elevation = (double*)take(n * 8);
  h.path_length = (double*)take(n * 8);
  h.normal = (double*)take(n * 24);
  h.rgba = (double*)take(n * 32);
  h.color_tag = (uint32_t*)take(n * 4);
  return h;
}
static size_t packed_bytes(size_t n) { return n * (5 * 8 + 24 + 32 + 4) + 8 * 256 + 256; }

// One frame of the Fast or Rectilinear generator into `dense` (device memory).  When `want_packed` (or whenever
// a pixel can hold several trace points) the packed trace points are left in c->d_packed and their offsets in
// ws.hit_offset.  Counters are NOT reset here.
static int run_core(atmrt_ctx* c, const Frame& f, Workspace& ws, const DensePlanes& dense, bool want_packed,
                    PackedHits* packed_out, uint64_t* n_hits_out) {
  hipStream_t s = c->stream;
  launch_resolve(f, ws, c->d_objects.as<ObjectDev>(), s);
  // phase events: [0..1] profile, [2..3] paths (stream2), [4..5] intersect / march, [5..6] finalize, [7..8] pack
  hipEvent_t* ev = c->ev;
  const bool fast = f.p.generator == ATMRT_GEN_FAST;
  const bool general = f.n_objects > 0; // scenes with objects: full get_single_pixel, count -> scan -> fill
  if (fast && general) {
    launch_fast_caches(f, ws, s, c->stream2, c->ev_fork, c->ev_join, ev);
    launch_close_count(f, ws, s);
    uint64_t cnt[4] = {0, 0, 0, 0};
    HIP_TRY(c, hipMemcpyAsync(cnt, ws.counters, sizeof cnt, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    HIP_TRY(c, c->d_clist.reserve((cnt[3] + 1) * sizeof(uint32_t)));
    ws.clist = c->d_clist.as<uint32_t>();
    launch_close_fill(f, ws, s);
    HIP_TRY(c, hipEventRecord(ev[4], s));
    launch_trace_count(f, ws, dense, s);
    HIP_TRY(c, hipEventRecord(ev[5], s));
    HIP_TRY(c, hipEventRecord(ev[6], s));
  } else if (general) {
    // Rectilinear with scene objects: the lean march first (it leaves the rays that can meet an object to the general tracer and
    // lists them), then the tracer over that list
    HIP_TRY(c, c->d_object_rays.reserve((size_t)f.wl * f.h * sizeof(uint32_t)));
    ws.object_rays = c->d_object_rays.as<uint32_t>();
    HIP_TRY(c, c->d_step_ctx.reserve((sizeof(Frame) + 255) / 256 * 256 + OBJECT_STEP_SINKS_MAX_BYTES));
    ws.step_ctx = c->d_step_ctx.as<char>();
    HIP_TRY(c, hipEventRecord(ev[4], s));
    launch_trace_count(f, ws, dense, s);
    uint64_t cnt[N_COUNTERS] = {};
    HIP_TRY(c, hipMemcpyAsync(cnt, ws.counters, sizeof cnt, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    launch_rect_trace_objects(f, ws, dense, cnt[11], s);
    HIP_TRY(c, hipEventRecord(ev[5], s));
    HIP_TRY(c, hipEventRecord(ev[6], s));
  } else if (fast) {
    c->scan_segments = launch_fast_pipeline(f, ws, dense, s, c->stream2, c->ev_fork, c->ev_seg, c->ev_scan, ev); // records ev[0..4]
    HIP_TRY(c, hipEventRecord(ev[5], s));
    if (f.opaque) launch_fast_finalize(f, ws, dense, s);
    HIP_TRY(c, hipEventRecord(ev[6], s));
  } else {
    HIP_TRY(c, hipEventRecord(ev[4], s));
    launch_rect_march(f, ws, dense, s, ev[5]);
    HIP_TRY(c, hipEventRecord(ev[6], s));
  }
  HIP_TRY(c, hipEventRecord(ev[7], s));
  PackedHits packed{};
  if (want_packed || !f.opaque) {
    uint64_t counters[N_COUNTERS] = {};
    launch_scan_counts(f, ws, dense.hit_count, s);
    HIP_TRY(c, hipMemcpyAsync(counters, ws.counters, sizeof counters, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
    uint64_t n_hits = counters[1];
    if (counters[6]) { // some step produced more trace points than StepHits keeps: the fill pass sorts those in place by `prop`
      HIP_TRY(c, c->d_step_prop.reserve((n_hits + 1) * sizeof(double)));
      ws.step_prop = c->d_step_prop.as<double>();
    }
    HIP_TRY(c, c->d_packed.reserve(packed_bytes(n_hits)));
    packed = carve_packed(c->d_packed.ptr, n_hits);
    if (f.opaque) {
      launch_pack_first_hits(f, ws, dense, packed, s);
    } else {
      HIP_TRY(c, c->d_list_step.reserve((n_hits + 1) * sizeof(uint32_t)));
      HIP_TRY(c, c->d_list_pixel.reserve((n_hits + 1) * sizeof(uint32_t)));
      ws.list_step = c->d_list_step.as<uint32_t>();
      ws.list_pixel = c->d_list_pixel.as<uint32_t>();
      if (f.p.generator == ATMRT_GEN_RECTILINEAR) {
        HIP_TRY(c, c->d_rect_rec.reserve(4 * (n_hits + 1) * sizeof(double)));
        ws.rect_rec = c->d_rect_rec.as<double>();
      }
      if (f.p.generator == ATMRT_GEN_RECTILINEAR) {
        ws.n_overflow = counters[3]; // pixels whose points did not fit the slots: marched a second time if the arena overflowed too
        ws.n_overflow_records = counters[13];
        HIP_TRY(c, c->d_overflow.reserve((ws.n_overflow + 1) * sizeof(uint32_t)));
        ws.overflow = c->d_overflow.as<uint32_t>();
        HIP_TRY(c, hipMemsetAsync(ws.counters + 3, 0, sizeof(uint64_t), s)); // now the gather kernel's list cursor
      }
      if (general) {
        launch_trace_fill(f, ws, n_hits, dense, packed, s);
      } else if (f.p.generator == ATMRT_GEN_RECTILINEAR) {
        launch_multi_fill(f, ws, n_hits, dense, packed, s);
      } else {
        launch_multi_fill_fast(f, ws, n_hits, dense, packed, s);
      }
    }
    if (n_hits_out) *n_hits_out = n_hits;
  }
  HIP_TRY(c, hipEventRecord(ev[8], s));
  if (packed_out) *packed_out = packed;
  return ATMRT_OK;
}

static int prepare_workspace(atmrt_ctx* c, const Frame& f, Workspace* ws);

// InterpolatingRectilinearGenerator::generate (interpolating_rectilinear.rs:110-162): ray table -> lattice steps ->
// lattice frame through the Fast pipeline -> 4-corner blend (count -> scan -> fill).
static int run_interpolating(atmrt_ctx* c, const Frame& f, Workspace& ws, const DensePlanes& dense, PackedHits* packed_out,
                             uint64_t* n_hits_out) {
  hipStream_t s = c->stream;
  const size_t W = f.p.width, H = f.p.height, npx = (size_t)f.wl * f.h;
  // gen_fov_data :453-522
  size_t bytes = 2 * W * H * 8 + (W + H) * 8 + npx * (4 + 4 + 8 + 8) + 64 + 4096;
  HIP_TRY(c, c->d_interp.reserve(bytes));
  InterpBuffers ib{};
  {
    char* p = c->d_interp.as<char>();
    auto take = [&](size_t b) {
      void* r = p;
      p += (b + 255) / 256 * 256;
      return r;
    };
    HIP_TRY(c, c->d_interp.reserve(bytes + 16 * 256));
    p = c->d_interp.as<char>();
    ib.dir = (double*)take(W * H * 8);
    ib.elev = (double*)take(W * H * 8);
    ib.colmin = (double*)take(W * 8);
    ib.rowmin = (double*)take(H * 8);
    ib.rem_e = (double*)take(npx * 8);
    ib.rem_d = (double*)take(npx * 8);
    ib.key_e = (int32_t*)take(npx * 4);
    ib.key_d = (int32_t*)take(npx * 4);
    ib.bounds = (int32_t*)take(16);
  }
  launch_fov_table(f, ib, s);
  std::vector<double> mins(W + H);
  HIP_TRY(c, hipMemcpyAsync(mins.data(), ib.colmin, W * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipMemcpyAsync(mins.data() + W, ib.rowmin, H * 8, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  double min_elev_step = INFINITY, min_dir_step = INFINITY; // .reduce(|| INFINITY, f64::min) * SCALE
  for (size_t x = 0; x < W; x++) min_elev_step = std::fmin(min_elev_step, mins[x]);
  for (size_t y = 0; y < H; y++) min_dir_step = std::fmin(min_dir_step, mins[W + y]);
  min_elev_step *= 1.5;
  min_dir_step *= 1.5;
  if (!(min_elev_step > 0.0) || !(min_dir_step > 0.0) || !std::isfinite(min_elev_step) || !std::isfinite(min_dir_step))
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "degenerate field of view for InterpolatingRectilinear");
  int32_t bounds[4] = {INT32_MAX, INT32_MIN, INT32_MAX, INT32_MIN};
  HIP_TRY(c, hipMemcpyAsync(ib.bounds, bounds, sizeof bounds, hipMemcpyHostToDevice, s));
  launch_lattice_keys(f, ib, min_elev_step, min_dir_step, s);
  HIP_TRY(c, hipMemcpyAsync(bounds, ib.bounds, sizeof bounds, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  const int64_t ne = (int64_t)bounds[1] + 1 - bounds[0] + 1, nd = (int64_t)bounds[3] + 1 - bounds[2] + 1;
  if (ne <= 0 || nd <= 0 || ne > 60000 || nd > 2000000 || ne * nd > (int64_t)1 << 31)
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "InterpolatingRectilinear lattice of %lld x %lld points is out of range",
                   (long long)nd, (long long)ne);
  // the lattice frame (Cache::get_pixel :80-107 for every lattice point of the bounding rectangle)
  Frame fl = f;
  fl.p.generator = ATMRT_GEN_FAST;
  fl.lattice = 1;
  fl.di0 = bounds[2];
  fl.ei0 = bounds[0];
  fl.dir_step = min_dir_step;
  fl.elev_step = min_elev_step;
  fl.c0 = 0;
  fl.wl = (int32_t)nd;
  fl.h = (int32_t)ne;
  const size_t nlat = (size_t)nd * ne;
  // run_core leaves its trace points in d_packed / d_hit_offset, and so must this frame.  The two pairs of buffers trade places
  // twice per frame — here, before the lattice workspace takes their addresses, and after the lattice pass — so that each pair serves the same role (lattice / image) in every frame
  // and keeps its capacity: with one swap the roles alternated and the smaller buffer was reallocated in the second frame.
  std::swap(c->d_packed, c->d_lat_packed);
  std::swap(c->d_hit_offset, c->d_lat_offset);
  Workspace wsl{};
  int rc = prepare_workspace(c, fl, &wsl);
  if (rc) return rc;
  HIP_TRY(c, c->d_px_steps.reserve(nlat * 4 + nlat + 256));
  wsl.px_steps = c->d_px_steps.as<uint32_t>();
  ib.referenced = reinterpret_cast<uint8_t*>(wsl.px_steps + nlat);
  HIP_TRY(c, hipMemsetAsync(ib.referenced, 0, nlat, s));
  HIP_TRY(c, c->d_lat_dense.reserve(dense_bytes(nlat)));
  DensePlanes ldense = carve_dense(c->d_lat_dense.ptr, nlat);
  PackedHits lpacked{};
  uint64_t lhits = 0;
  if ((rc = run_core(c, fl, wsl, ldense, true, &lpacked, &lhits))) return rc;
  std::swap(c->d_packed, c->d_lat_packed);   // keep the lattice result; the image gets the other pair
  std::swap(c->d_hit_offset, c->d_lat_offset);
  LatticeResult lr{};
  lr.hit_count = ldense.hit_count;
  lr.hit_offset = c->d_lat_offset.as<uint64_t>();
  lr.azimuth = ldense.azimuth;
  lr.elevation_angle = ldense.elevation_angle;
  lr.hits = lpacked;
  lr.px_steps = wsl.px_steps;
  lr.nd = (int32_t)nd;
  lr.ne = (int32_t)ne;
  // blend: count -> scan -> fill
  if ((rc = prepare_workspace(c, f, &ws))) return rc;
  HIP_TRY(c, hipMemsetAsync(ws.counters, 0, sizeof(uint64_t), s)); // ray-steps: only referenced lattice pixels count
  PackedHits none{};
  Frame fb = f;
  fb.di0 = fl.di0;
  fb.ei0 = fl.ei0;
  launch_interp_blend(fb, ws, ib, lr, false, dense, none, s);
  launch_scan_counts(f, ws, dense.hit_count, s);
  uint64_t counters[N_COUNTERS] = {};
  HIP_TRY(c, hipMemcpyAsync(counters, ws.counters, sizeof counters, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  BlendArena arena{};
  if (counters[7]) { // pixels whose four corners hold more points than the in-register member list: blended over an HBM arena
    const size_t n = (size_t)counters[8];
    HIP_TRY(c, c->d_blend_arena.reserve(n * (8 + 8 + 4 + 1 + 1) + 4 * 256));
    char* a = c->d_blend_arena.as<char>();
    auto take = [&](size_t b) {
      void* r = a;
      a += (b + 255) / 256 * 256;
      return r;
    };
    arena.k = (uint64_t*)take(n * 8);
    arena.dist = (double*)take(n * 8);
    arena.group = (uint32_t*)take(n * 4);
    arena.corner = (uint8_t*)take(n);
    arena.tag = (uint8_t*)take(n);
    launch_interp_blend_big(fb, ws, ib, lr, false, dense, none, arena, s);
    launch_scan_counts(f, ws, dense.hit_count, s); // again, now that every pixel has its count
    HIP_TRY(c, hipMemcpyAsync(counters, ws.counters, sizeof counters, hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));
  }
  uint64_t n_hits = counters[1];
  HIP_TRY(c, c->d_packed.reserve(packed_bytes(n_hits)));
  PackedHits packed = carve_packed(c->d_packed.ptr, n_hits);
  launch_interp_blend(fb, ws, ib, lr, true, dense, packed, s);
  if (arena.k) launch_interp_blend_big(fb, ws, ib, lr, true, dense, packed, arena, s);
  launch_interp_finish(f, ws, ib, lr, dense, packed, s);
  if (packed_out) *packed_out = packed;
  if (n_hits_out) *n_hits_out = n_hits;
  return ATMRT_OK;
}

// Runs the generator named in params.  `dense` must be device memory.  When `want_packed`, the
// packed trace points are left in c->d_packed (n_hits of them).
static int run_generator(atmrt_ctx* c, const Frame& f, Workspace& ws, const DensePlanes& dense, bool want_packed,
                         PackedHits* packed_out, uint64_t* n_hits_out, uint64_t* ray_steps_out, double* ms_out) {
  hipStream_t s = c->stream;
  hipEvent_t* ev = c->ev;
  const bool fast = f.p.generator == ATMRT_GEN_FAST;
  // the buffers of the previous frame are about to be reused (or reallocated): until this frame has succeeded there is nothing
  // atmrt_draw_image / atmrt_last_hits_device may touch
  c->last_valid = false;
  c->stats = atmrt_frame_stats_t{};
  HIP_TRY(c, hipEventRecord(c->ev_t0, s));
  c->scan_segments = 0;
  HIP_TRY(c, hipMemsetAsync(ws.counters, 0, N_COUNTERS * sizeof(uint64_t), s));
  PackedHits packed{};
  int rc;
  if (f.p.generator == ATMRT_GEN_INTERPOLATING_RECTILINEAR) rc = run_interpolating(c, f, ws, dense, &packed, n_hits_out);
  else rc = run_core(c, f, ws, dense, want_packed, &packed, n_hits_out);
  if (rc) return rc;
  if (c->inject_failure) { // test hook: the frame's kernels have run and its buffers have been rewritten; now fail
    c->inject_failure = false;
    (void)hipStreamSynchronize(s);
    return c->fail(ATMRT_ERR_HIP, "failure injected by atmrt_debug_fail_next_frame");
  }
  uint64_t counters[N_COUNTERS] = {};
  HIP_TRY(c, hipEventRecord(c->ev_t1, s));
  HIP_TRY(c, hipMemcpyAsync(counters, ws.counters, sizeof counters, hipMemcpyDeviceToHost, s));
  HIP_TRY(c, hipStreamSynchronize(s));
  HIP_TRY(c, hipGetLastError());
  float ms = 0.f;
  HIP_TRY(c, hipEventElapsedTime(&ms, c->ev_t0, c->ev_t1));
  {
    atmrt_timings_t t{};
    float v = 0.f;
    t.total_ms = ms;
    if (f.p.generator != ATMRT_GEN_RECTILINEAR) {
      HIP_TRY(c, hipEventElapsedTime(&v, ev[0], ev[1]));
      t.profile_ms = v;
      HIP_TRY(c, hipEventElapsedTime(&v, ev[2], ev[3]));
      t.paths_ms = v;
      HIP_TRY(c, hipEventElapsedTime(&v, ev[4], ev[5]));
      t.intersect_ms = v;
      if (c->scan_segments) { // pipelined frame: the scan's own time, without the waits for the path segments
        t.intersect_ms = 0.0;
        for (int k = 0; k < c->scan_segments; k++) {
          HIP_TRY(c, hipEventElapsedTime(&v, c->ev_scan[2 * k], c->ev_scan[2 * k + 1]));
          t.intersect_ms += v;
        }
      }
    } else {
      HIP_TRY(c, hipEventElapsedTime(&v, ev[4], ev[5]));
      t.march_ms = v;
    }
    HIP_TRY(c, hipEventElapsedTime(&v, ev[5], ev[6]));
    t.finalize_ms = v;
    HIP_TRY(c, hipEventElapsedTime(&v, ev[7], ev[8]));
    t.pack_ms = v;
    t.ray_steps = counters[0];
    t.n_hits = counters[1];
    c->timings = t;
  }
  (void)fast;
  if (counters[12])
    return c->fail(ATMRT_ERR_HIP, "the time-sliced march left %llu of its ray groups unfinished", (unsigned long long)counters[12] - 1);
  c->stats.unlisted_rays = counters[4];
  c->stats.unlisted_columns = counters[5];
  c->stats.big_steps = counters[6];
  c->stats.big_blend_pixels += counters[7];
  c->stats.retraced_pixels += ws.n_overflow;
  c->stats.terrain_lookups = counters[10];
  c->stats.object_rays = counters[11];
  c->stats.object_steps = counters[14];
  if (ms_out) *ms_out = ms;
  if (ray_steps_out) *ray_steps_out = counters[0];
  if (packed_out) *packed_out = packed;
  c->last_valid = true;
  c->last_packed = want_packed || !f.opaque || f.p.generator == ATMRT_GEN_INTERPOLATING_RECTILINEAR;
  c->last_npx = (size_t)f.wl * f.h;
  c->last_wl = f.wl;
  c->last_h = f.h;
  c->last_c0 = f.c0;
  c->last_alpha = f.p.terrain_alpha;
  c->last_dense = dense;
  c->last_hits = packed;
  c->last_offset = ws.hit_offset;
  c->last_nhits = c->last_packed ? counters[1] : 0;
  return ATMRT_OK;
}

// ---------------------------------------------------------------------------------------------
// the path
// ---------------------------------------------------------------------------------------------
// Host memory of atmrt_result_t: ONE page-locked block per result (device-to-host copies run at PCIe speed into it; through
// pageable memory the 0.7 GB of a headline frame cost 55 ms, eight times the Fast generator's device time), carved into the
// arrays with 256-byte alignment.  Pinning is slow, so freed blocks are kept (at most two, process-wide) and reused by later
// frames of a similar size.  If pinned memory cannot be had the block is ordinary malloc memory.
namespace {
struct HostBlocks {
  std::mutex m;
  struct Blk {
    void* p;
    size_t cap;
    bool pinned;
  };
  std::vector<Blk> live, spare;
  void* take(size_t bytes) {
    std::lock_guard<std::mutex> g(m);
    for (size_t i = 0; i < spare.size(); i++)
      if (spare[i].cap >= bytes && spare[i].cap / 2 <= bytes) {
        Blk b = spare[i];
        spare.erase(spare.begin() + (long)i);
        live.push_back(b);
        return b.p;
      }
    Blk b{nullptr, bytes + bytes / 16, true};
    if (hipHostMalloc(&b.p, b.cap, hipHostMallocDefault) != hipSuccess || !b.p) {
      (void)hipGetLastError();
      b = Blk{malloc(bytes), bytes, false};
      if (!b.p) return nullptr;
    }
    live.push_back(b);
    return b.p;
  }
  void give(void* p) {
    std::lock_guard<std::mutex> g(m);
    for (size_t i = 0; i < live.size(); i++)
      if (live[i].p == p) {
        Blk b = live[i];
        live.erase(live.begin() + (long)i);
        if (!b.pinned) {
          free(b.p);
          return;
        }
        spare.push_back(b);
        if (spare.size() > 2) {
          (void)hipHostFree(spare.front().p);
          spare.erase(spare.begin());
        }
        return;
      }
  }
};
HostBlocks g_host_blocks;
} // namespace

extern "C" int atmrt_internal_result_alloc(atmrt_result_t* out, uint32_t width, uint32_t height, uint64_t n_hits) {
  memset(out, 0, sizeof *out);
  const size_t npx = (size_t)width * height;
  const size_t nh = n_hits ? n_hits : 1, np1 = npx ? npx : 1;
  auto pad = [](size_t b) { return (b + 255) / 256 * 256; };
  const size_t total = 3 * pad(np1 * 8) + pad(np1 * 4) + 5 * pad(nh * 8) + pad(nh * 24) + pad(nh * 4) + pad(nh * 32);
  char* base = static_cast<char*>(g_host_blocks.take(total));
  if (!base) return -1;
  auto carve = [&](size_t bytes) {
    void* r = base;
    base += pad(bytes);
    return r;
  };
  out->width = width;
  out->height = height;
  out->n_pixels = npx;
  out->n_hits = n_hits;
  out->azimuth = (double*)carve(np1 * 8); // first: atmrt_result_free returns the block by this pointer
  out->elevation_angle = (double*)carve(np1 * 8);
  out->hit_offset = (uint64_t*)carve(np1 * 8);
  out->hit_count = (uint32_t*)carve(np1 * 4);
  out->lat = (double*)carve(nh * 8);
  out->lon = (double*)carve(nh * 8);
  out->distance = (double*)carve(nh * 8);
  out->elevation = (double*)carve(nh * 8);
  out->path_length = (double*)carve(nh * 8);
  out->normal = (double*)carve(nh * 24);
  out->rgba = (double*)carve(nh * 32);
  out->color_tag = (uint32_t*)carve(nh * 4);
  return 0;
}

extern "C" void atmrt_result_free(atmrt_result_t* r) {
  if (!r) return;
  if (r->azimuth) g_host_blocks.give(r->azimuth); // the first array is the base of the block
  memset(r, 0, sizeof *r);
}

// One frame of this context's tile, everything left in HBM (shared by atmrt_generate_device and the multi-device paths).
int atmrt::api_generate_tile(atmrt_ctx* c, const DensePlanes* dense_in, bool want_packed, uint64_t* n_hits, uint64_t* ray_steps,
                             double* device_ms) {
  Frame f;
  int rc = prepare_frame(c, &f);
  if (rc) return rc;
  Workspace ws{};
  if ((rc = prepare_workspace(c, f, &ws))) return rc;
  DensePlanes dense;
  if (dense_in) {
    dense = *dense_in;
  } else {
    const size_t npx = (size_t)f.wl * f.h;
    HIP_TRY(c, c->d_dense.reserve(dense_bytes(npx)));
    dense = carve_dense(c->d_dense.ptr, npx);
  }
  uint64_t nh = 0;
  rc = run_generator(c, f, ws, dense, want_packed, nullptr, &nh, ray_steps, device_ms);
  if (n_hits) *n_hits = nh;
  return rc;
}

extern "C" int atmrt_generate(atmrt_ctx* c, atmrt_result_t* out) {
  if (!c || !out) return ATMRT_ERR_INVALID_ARGUMENT;
  memset(out, 0, sizeof *out);
  if (c->multi) return multi_generate(c, out); // every device its pixel-column tile, straight into the one [H][W] block
  Frame f;
  int rc = prepare_frame(c, &f);
  if (rc) return rc;
  Workspace ws{};
  if ((rc = prepare_workspace(c, f, &ws))) return rc;
  size_t npx = (size_t)f.wl * f.h;
  HIP_TRY(c, c->d_dense.reserve(dense_bytes(npx)));
  DensePlanes dense = carve_dense(c->d_dense.ptr, npx);
  PackedHits packed{};
  uint64_t n_hits = 0, steps = 0;
  double ms = 0;
  if ((rc = run_generator(c, f, ws, dense, true, &packed, &n_hits, &steps, &ms))) return rc;

  if (atmrt_internal_result_alloc(out, (uint32_t)f.wl, (uint32_t)f.h, n_hits))
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "out of host memory for %zu pixels / %llu hits", npx, (unsigned long long)n_hits);
  out->ray_steps = steps;
  out->device_ms = ms;
  hipStream_t s = c->stream;
  auto d2h = [&](void* dst, const void* src, size_t bytes) { return hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, s); };
  hipError_t e = d2h(out->azimuth, dense.azimuth, npx * 8);
  if (e == hipSuccess) e = d2h(out->elevation_angle, dense.elevation_angle, npx * 8);
  if (e == hipSuccess) e = d2h(out->hit_count, dense.hit_count, npx * 4);
  if (e == hipSuccess) e = d2h(out->hit_offset, ws.hit_offset, npx * 8);
  if (n_hits) {
    if (e == hipSuccess) e = d2h(out->lat, packed.lat, n_hits * 8);
    if (e == hipSuccess) e = d2h(out->lon, packed.lon, n_hits * 8);
    if (e == hipSuccess) e = d2h(out->distance, packed.distance, n_hits * 8);
    if (e == hipSuccess) e = d2h(out->elevation, packed.elevation, n_hits * 8);
    if (e == hipSuccess) e = d2h(out->path_length, packed.path_length, n_hits * 8);
    if (e == hipSuccess) e = d2h(out->normal, packed.normal, n_hits * 24);
    if (e == hipSuccess) e = d2h(out->color_tag, packed.color_tag, n_hits * 4);
    if (e == hipSuccess) e = d2h(out->rgba, packed.rgba, n_hits * 32);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(s);
  if (e != hipSuccess) {
    atmrt_result_free(out);
    return c->fail(ATMRT_ERR_HIP, "copying the result to the host: %s", hipGetErrorString(e));
  }
  return ATMRT_OK;
}

extern "C" int atmrt_generate_device(atmrt_ctx* c, const atmrt_device_planes_t* planes, uint64_t* ray_steps,
                                     double* device_ms) {
  if (!c || !planes) return ATMRT_ERR_INVALID_ARGUMENT;
  if (c->multi) return c->fail(ATMRT_ERR_STATE, "a multi-device context leaves its frame in HBM through atmrt_generate_image_device");
  if (!planes->azimuth || !planes->elevation_angle || !planes->hit_count || !planes->lat || !planes->lon ||
      !planes->distance || !planes->elevation || !planes->path_length || !planes->normal)
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "every plane pointer must be a device allocation");
  Frame f;
  int rc = prepare_frame(c, &f);
  if (rc) return rc;
  Workspace ws{};
  if ((rc = prepare_workspace(c, f, &ws))) return rc;
  DensePlanes dense;
  dense.azimuth = planes->azimuth;
  dense.elevation_angle = planes->elevation_angle;
  dense.hit_count = planes->hit_count;
  dense.lat = planes->lat;
  dense.lon = planes->lon;
  dense.distance = planes->distance;
  dense.elevation = planes->elevation;
  dense.path_length = planes->path_length;
  dense.normal = planes->normal;
  uint64_t n_hits = 0;
  return run_generator(c, f, ws, dense, false, nullptr, &n_hits, ray_steps, device_ms);
}

extern "C" int atmrt_last_hits_device(atmrt_ctx* c, const atmrt_device_hits_t* dst, uint64_t* n_hits) {
  if (!c) return ATMRT_ERR_INVALID_ARGUMENT;
  if (c->multi) return c->fail(ATMRT_ERR_STATE, "a multi-device context hands its lists over through atmrt_image_hits_device");
  if (!c->last_valid) return c->fail(ATMRT_ERR_STATE, "atmrt_last_hits_device needs a frame: call atmrt_generate_device first");
  if (!c->last_packed)
    return c->fail(ATMRT_ERR_STATE, "the last frame holds first-hit planes only (opaque scene through atmrt_generate_device): "
                                    "its trace points are the planes themselves");
  const uint64_t n = c->last_nhits;
  if (n_hits) *n_hits = n;
  if (!dst) return ATMRT_OK; // size query
  if (dst->capacity < n)
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "capacity %llu is less than the %llu trace points of the frame",
                   (unsigned long long)dst->capacity, (unsigned long long)n);
  if (!dst->hit_offset || !dst->lat || !dst->lon || !dst->distance || !dst->elevation || !dst->path_length || !dst->normal ||
      !dst->color_tag || !dst->rgba)
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "every array pointer must be a device allocation");
  HIP_TRY(c, hipSetDevice(c->device));
  hipStream_t s = c->stream;
  const PackedHits& h = c->last_hits;
  auto d2d = [&](void* to, const void* from, size_t bytes) { return bytes ? hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, s) : hipSuccess; };
  HIP_TRY(c, d2d(dst->hit_offset, c->last_offset, c->last_npx * 8));
  HIP_TRY(c, d2d(dst->lat, h.lat, n * 8));
  HIP_TRY(c, d2d(dst->lon, h.lon, n * 8));
  HIP_TRY(c, d2d(dst->distance, h.distance, n * 8));
  HIP_TRY(c, d2d(dst->elevation, h.elevation, n * 8));
  HIP_TRY(c, d2d(dst->path_length, h.path_length, n * 8));
  HIP_TRY(c, d2d(dst->normal, h.normal, n * 24));
  HIP_TRY(c, d2d(dst->color_tag, h.color_tag, n * 4));
  HIP_TRY(c, d2d(dst->rgba, h.rgba, n * 32));
  HIP_TRY(c, hipStreamSynchronize(s));
  return ATMRT_OK;
}

extern "C" int atmrt_last_timings(atmrt_ctx* c, atmrt_timings_t* out) {
  if (!c || !out) return ATMRT_ERR_INVALID_ARGUMENT;
  if (c->multi) return multi_last_timings(c, out);
  *out = c->timings;
  return ATMRT_OK;
}

extern "C" int atmrt_debug_fail_next_frame(atmrt_ctx* c) {
  if (!c) return ATMRT_ERR_INVALID_ARGUMENT;
  if (c->multi) return atmrt_debug_fail_next_frame(multi_child(c, multi_size(c) - 1));
  c->inject_failure = true;
  return ATMRT_OK;
}

extern "C" int atmrt_debug_march_plan(int32_t width, int32_t height, int32_t samples, int32_t n_objects, uint64_t out[6]) {
  if (!out || width < 0 || height < 0 || samples < 0 || n_objects < 0) return ATMRT_ERR_INVALID_ARGUMENT;
  Frame f{};
  f.p.generator = ATMRT_GEN_RECTILINEAR;
  f.wl = width;
  f.h = height;
  f.n_t = samples;
  f.n_objects = n_objects;
  f.opaque = n_objects == 0; // (the plan of a frame over opaque terrain: translucent terrain is not sliced, like scenes with objects)
  SliceLayout L{};
  const bool sliced = march_slice_layout(f, L);
  out[0] = sliced ? 1 : 0;
  out[1] = sliced ? L.n_groups : 0;
  out[2] = sliced ? L.cap : 0;
  out[3] = sliced ? L.bytes : 0;
  out[4] = sliced ? L.slices_after : 0;
  out[5] = MARCH_SLICE_STEPS;
  return ATMRT_OK;
}

extern "C" int atmrt_last_stats(atmrt_ctx* c, atmrt_frame_stats_t* out) {
  if (!c || !out) return ATMRT_ERR_INVALID_ARGUMENT;
  if (c->multi) return multi_last_stats(c, out);
  *out = c->stats;
  return ATMRT_OK;
}

// ---------------------------------------------------------------------------------------------
// renderer compositing + colouring
// ---------------------------------------------------------------------------------------------
extern "C" int atmrt_coloring_from_conf(const atmrt_params_t* params, int32_t kind, double water_level, double ambient_light,
                                        double light_zenith_angle, double light_dir, int32_t palette, int32_t has_fog,
                                        double fog_distance, atmrt_coloring_t* out) {
  if (!params || !out) return ATMRT_ERR_INVALID_ARGUMENT;
  if (palette != ATMRT_PALETTE_LEGACY && palette != ATMRT_PALETTE_IMPROVED) return ATMRT_ERR_INVALID_ARGUMENT;
  return coloring_from_conf(*params, kind, water_level, ambient_light, light_zenith_angle, light_dir, palette, has_fog,
                            fog_distance, *out)
             ? ATMRT_ERR_INVALID_ARGUMENT
             : ATMRT_OK;
}

extern "C" int atmrt_draw_image_device(atmrt_ctx* c, const atmrt_coloring_t* coloring, uint8_t* rgb_device) {
  if (!c || !coloring || !rgb_device) return ATMRT_ERR_INVALID_ARGUMENT;
  if (c->multi) return c->fail(ATMRT_ERR_STATE, "a multi-device context draws through atmrt_draw_image (host image) or atmrt_draw_image_gathered_device");
  if (!c->last_valid) return c->fail(ATMRT_ERR_STATE, "atmrt_draw_image needs a frame: call atmrt_generate first");
  if (coloring->kind != ATMRT_COLORING_SIMPLE && coloring->kind != ATMRT_COLORING_SHADING)
    return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "unknown coloring kind %d", coloring->kind);
  if (coloring->has_fog && !(coloring->fog_distance > 0.0)) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "fog_distance must be positive");
  HIP_TRY(c, hipSetDevice(c->device));
  launch_draw_image(c->last_npx, *coloring, c->last_alpha, c->last_packed, c->last_dense.hit_count, c->last_offset, c->last_hits,
                    c->last_dense, rgb_device, c->stream);
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  return ATMRT_OK;
}

extern "C" int atmrt_draw_image(atmrt_ctx* c, const atmrt_coloring_t* coloring, uint8_t* rgb) {
  if (!c || !coloring || !rgb) return ATMRT_ERR_INVALID_ARGUMENT;
  if (c->multi) return multi_draw_image(c, coloring, rgb);
  if (!c->last_valid) return c->fail(ATMRT_ERR_STATE, "atmrt_draw_image needs a frame: call atmrt_generate first");
  HIP_TRY(c, hipSetDevice(c->device)); // the staging buffer must live on this context's device
  HIP_TRY(c, c->d_io.reserve(3 * c->last_npx + 256));
  int rc = atmrt_draw_image_device(c, coloring, c->d_io.as<uint8_t>());
  if (rc) return rc;
  HIP_TRY(c, hipMemcpy(rgb, c->d_io.ptr, 3 * c->last_npx, hipMemcpyDeviceToHost));
  return ATMRT_OK;
}

// ---------------------------------------------------------------------------------------------
// harnesses: host arrays in, host arrays out (staged through one device buffer)
// ---------------------------------------------------------------------------------------------
// the diagnostic entry points of a multi-device context run on its first device
#define FORWARD_TO_FIRST_DEVICE(c, call) \
  do {                                   \
    if ((c) && (c)->multi) {             \
      atmrt_ctx* k_ = multi_child((c), 0); \
      int rc_ = (call);                  \
      if (rc_) (c)->error = k_->error;   \
      return rc_;                        \
    }                                    \
  } while (0)

static int harness_frame(atmrt_ctx* c, Frame* f) {
  if (!c->have_params) {
    atmrt_params_t p;
    atmrt_params_default(&p);
    int rc = atmrt_set_params(c, &p);
    if (rc) return rc;
  }
  return prepare_frame(c, f);
}

extern "C" int atmrt_terrain_get_elev(atmrt_ctx* c, size_t n, const double* lat, const double* lon, double* elev,
                                      uint8_t* valid) {
  if (!c || (n && (!lat || !lon || !elev || !valid))) return ATMRT_ERR_INVALID_ARGUMENT;
  FORWARD_TO_FIRST_DEVICE(c, atmrt_terrain_get_elev(k_, n, lat, lon, elev, valid));
  Frame f;
  int rc = harness_frame(c, &f);
  if (rc) return rc;
  if (!n) return ATMRT_OK;
  HIP_TRY(c, c->d_io.reserve(n * 25 + 1024));
  double* d_lat = c->d_io.as<double>();
  double* d_lon = d_lat + n;
  double* d_elev = d_lon + n;
  uint8_t* d_valid = reinterpret_cast<uint8_t*>(d_elev + n);
  HIP_TRY(c, hipMemcpyAsync(d_lat, lat, n * 8, hipMemcpyHostToDevice, c->stream));
  HIP_TRY(c, hipMemcpyAsync(d_lon, lon, n * 8, hipMemcpyHostToDevice, c->stream));
  launch_get_elev(f, n, d_lat, d_lon, d_elev, d_valid, c->stream);
  HIP_TRY(c, hipMemcpyAsync(elev, d_elev, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(valid, d_valid, n, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  return ATMRT_OK;
}

extern "C" int atmrt_ray_paths(atmrt_ctx* c, double h0, size_t n_angles, const double* angles_deg, int32_t straight,
                               double step, size_t n_steps, double* x, double* h) {
  if (!c || (n_angles && (!angles_deg || !x || !h))) return ATMRT_ERR_INVALID_ARGUMENT;
  FORWARD_TO_FIRST_DEVICE(c, atmrt_ray_paths(k_, h0, n_angles, angles_deg, straight, step, n_steps, x, h));
  if (!(step > 0.0)) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "step must be positive"); // ray_path.rs:53
  if (n_steps > 50000000) return c->fail(ATMRT_ERR_INVALID_ARGUMENT, "too many steps");
  Frame f;
  int rc = harness_frame(c, &f);
  if (rc) return rc;
  if (!n_angles) return ATMRT_OK;
  size_t m = n_angles * (n_steps + 1);
  HIP_TRY(c, c->d_io.reserve((n_angles + 2 * m) * 8));
  double* d_ang = c->d_io.as<double>();
  double* d_x = d_ang + n_angles;
  double* d_h = d_x + m;
  HIP_TRY(c, hipMemcpyAsync(d_ang, angles_deg, n_angles * 8, hipMemcpyHostToDevice, c->stream));
  launch_ray_paths(f, h0, n_angles, d_ang, straight, step, n_steps, d_x, d_h, c->stream);
  HIP_TRY(c, hipMemcpyAsync(x, d_x, m * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(h, d_h, m * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  return ATMRT_OK;
}

extern "C" int atmrt_atmosphere_sample(atmrt_ctx* c, size_t n, const double* altitude, double* temperature,
                                       double* pressure, double* n_index, double* dn_dh) {
  if (!c || (n && (!altitude || !temperature || !pressure || !n_index || !dn_dh))) return ATMRT_ERR_INVALID_ARGUMENT;
  FORWARD_TO_FIRST_DEVICE(c, atmrt_atmosphere_sample(k_, n, altitude, temperature, pressure, n_index, dn_dh));
  Frame f;
  int rc = harness_frame(c, &f);
  if (rc) return rc;
  if (!n) return ATMRT_OK;
  HIP_TRY(c, c->d_io.reserve(5 * n * 8));
  double* d = c->d_io.as<double>();
  HIP_TRY(c, hipMemcpyAsync(d, altitude, n * 8, hipMemcpyHostToDevice, c->stream));
  launch_atm_sample(f, n, d, d + n, d + 2 * n, d + 3 * n, d + 4 * n, c->stream);
  HIP_TRY(c, hipMemcpyAsync(temperature, d + n, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(pressure, d + 2 * n, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(n_index, d + 3 * n, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(dn_dh, d + 4 * n, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  return ATMRT_OK;
}

extern "C" int atmrt_coords_at_dist(atmrt_ctx* c, double lat0, double lon0, double dir_deg, size_t n,
                                    const double* dist, double* lat, double* lon) {
  if (!c || (n && (!dist || !lat || !lon))) return ATMRT_ERR_INVALID_ARGUMENT;
  FORWARD_TO_FIRST_DEVICE(c, atmrt_coords_at_dist(k_, lat0, lon0, dir_deg, n, dist, lat, lon));
  Frame f;
  int rc = harness_frame(c, &f);
  if (rc) return rc;
  if (!n) return ATMRT_OK;
  HIP_TRY(c, c->d_io.reserve(3 * n * 8));
  double* d = c->d_io.as<double>();
  HIP_TRY(c, hipMemcpyAsync(d, dist, n * 8, hipMemcpyHostToDevice, c->stream));
  launch_coords_at_dist(f, lat0, lon0, dir_deg, n, d, d + n, d + 2 * n, c->stream);
  HIP_TRY(c, hipMemcpyAsync(lat, d + n, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipMemcpyAsync(lon, d + 2 * n, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  return ATMRT_OK;
}

extern "C" int atmrt_math_probe(atmrt_ctx* c, int32_t op, size_t n, const double* a, const double* b, double* out0, double* out1) {
  if (!c || (n && (!a || !out0)) || op < 0 || op > ATMRT_PROBE_POW3_SHARED) return ATMRT_ERR_INVALID_ARGUMENT;
  FORWARD_TO_FIRST_DEVICE(c, atmrt_math_probe(k_, op, n, a, b, out0, out1));
  if (!n) return ATMRT_OK;
  HIP_TRY(c, hipSetDevice(c->device));
  HIP_TRY(c, c->d_io.reserve(4 * n * 8));
  double* d = c->d_io.as<double>();
  HIP_TRY(c, hipMemcpyAsync(d, a, n * 8, hipMemcpyHostToDevice, c->stream));
  if (b) HIP_TRY(c, hipMemcpyAsync(d + n, b, n * 8, hipMemcpyHostToDevice, c->stream));
  launch_math_probe(op, n, d, b ? d + n : nullptr, d + 2 * n, d + 3 * n, c->stream);
  HIP_TRY(c, hipMemcpyAsync(out0, d + 2 * n, n * 8, hipMemcpyDeviceToHost, c->stream));
  if (out1) HIP_TRY(c, hipMemcpyAsync(out1, d + 3 * n, n * 8, hipMemcpyDeviceToHost, c->stream));
  HIP_TRY(c, hipStreamSynchronize(c->stream));
  HIP_TRY(c, hipGetLastError());
  return ATMRT_OK;
}

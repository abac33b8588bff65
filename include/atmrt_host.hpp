// atmrt_host.hpp — header-only C++17 host mirror of the reference's generator interface above the C ABI
// (include/atmrt.h).  Names, argument meaning and error behaviour follow the Rust reference:
//
//   reference (Rust)                                           here (namespace atmrt_host)
//   ---------------------------------------------------------  -------------------------------------------
//   Terrain::from_folder(path) / get_elev   terrain/mod.rs:66-126   Terrain::from_folder / get_elev
//   Params (view, model, env, straight_rays, simulation_step,       Params
//           output, scene)                   params.rs:496-505
//   trait Generator { fn generate(&self) -> Vec<Vec<ResultPixel>> }  struct Generator { virtual generate() const }
//   FastGenerator::new(&params, &terrain, start)  fast.rs:102-108     FastGenerator(params, terrain)
//   RectilinearGenerator / InterpolatingRectilinearGenerator         same names
//   match params.output.generator  generator/mod.rs:72-78            make_generator(params, terrain)
//   ResultPixel / TracePoint / PixelColor  generators/mod.rs:13-80   same names
//   panic!/expect on bad input                                       throws std::runtime_error with the library's message
//
//   (one GPU in the reference's world)                               Terrain(std::vector<int> devices): the same frame cut into
//                                                                    pixel-column tiles over several GPUs, inside the library
//
// Link with -latmrt.  There is no CPU path: constructing a Terrain without a gfx950 device throws.
#pragma once

#include <array>
#include <memory>
#include <optional>
#include <stdexcept>
#include <string>
#include <vector>

#include "atmrt.h"

namespace atmrt_host {

struct Error : std::runtime_error {
  int status;
  Error(int s, const std::string& m) : std::runtime_error(m), status(s) {}
};

struct Altitude { // params.rs:17-21
  enum Kind { Absolute = ATMRT_ALT_ABSOLUTE, Relative = ATMRT_ALT_RELATIVE } kind = Relative;
  double value = 1.0;
};
struct Position { // params.rs:32-40
  double latitude = 0.0, longitude = 0.0;
  Altitude altitude;
};
struct Frame { // params.rs:145-155
  double direction = 0.0, tilt = 0.0, fov = 30.0, max_distance = 150000.0;
};
struct EarthModel { // earth_model/mod.rs:18-28
  atmrt_earth_model_t pod{ATMRT_EARTH_SPHERICAL, 0, 6371000.0, 0.0, 0.0};
  static EarthModel SimpleSphere() { return make(ATMRT_EARTH_SIMPLE_SPHERE); }
  static EarthModel Spherical(double radius) { EarthModel m = make(ATMRT_EARTH_SPHERICAL); m.pod.radius = radius; return m; }
  static EarthModel Ellipsoid(double a, double b) { EarthModel m = make(ATMRT_EARTH_ELLIPSOID); m.pod.a = a; m.pod.b = b; return m; }
  static EarthModel Wgs84() { return make(ATMRT_EARTH_WGS84); }
  static EarthModel AzimuthalEquidistant() { return make(ATMRT_EARTH_AZIMUTHAL_EQUIDISTANT); }
  static EarthModel FlatDistorted() { return make(ATMRT_EARTH_FLAT_DISTORTED); }
  static EarthModel ObserverAe(double proj_radius) { EarthModel m = make(ATMRT_EARTH_OBSERVER_AE); m.pod.radius = proj_radius; return m; }
  static EarthModel SimpleObserverAe() { return make(ATMRT_EARTH_SIMPLE_OBSERVER_AE); }
 private:
  static EarthModel make(int kind) { EarthModel m; m.pod = atmrt_earth_model_t{kind, 0, 0.0, 0.0, 0.0}; return m; }
};
enum class GeneratorDef { Fast = ATMRT_GEN_FAST, InterpolatingRectilinear = ATMRT_GEN_INTERPOLATING_RECTILINEAR, Rectilinear = ATMRT_GEN_RECTILINEAR };

struct Color { double r = 0, g = 0, b = 0, a = 1.0; }; // object/mod.rs:133-146
struct Object { // ConfObject after into_shape, object/mod.rs:42-75,156-161
  atmrt_object_t pod{};
  std::vector<uint8_t> texture; // RGBA8, top row first
  static Object Frustum(Position p, double r1, double r2, double height, Color c) {
    Object o; o.init(p); o.pod.kind = ATMRT_OBJ_FRUSTUM; o.pod.r1 = r1; o.pod.r2 = r2; o.pod.height = height;
    o.pod.color[0] = c.r; o.pod.color[1] = c.g; o.pod.color[2] = c.b; o.pod.color[3] = c.a; return o;
  }
  static Object Cylinder(Position p, double radius, double height, Color c) { return Frustum(p, radius, radius, height, c); }
  static Object Cone(Position p, double radius, double height, Color c) { return Frustum(p, radius, 0.0, height, c); }
  static Object Billboard(Position p, double width, double height, std::vector<uint8_t> rgba, uint32_t tex_w, uint32_t tex_h) {
    Object o; o.init(p); o.pod.kind = ATMRT_OBJ_BILLBOARD; o.pod.width = width; o.pod.height = height;
    o.texture = std::move(rgba); o.pod.texture_width = tex_w; o.pod.texture_height = tex_h; return o;
  }
 private:
  void init(Position p) { pod.position = atmrt_position_t{p.latitude, p.longitude, (int32_t)p.altitude.kind, 0, p.altitude.value}; }
};

struct Params { // params.rs:496-505 (the fields that reach the generators); defaults = Config::default :481-494
  Position position;
  Frame frame;
  EarthModel model;
  double wavelength = 530e-9;
  bool straight_rays = false;
  double simulation_step = 50.0;
  uint16_t width = 640, height = 480;
  GeneratorDef generator = GeneratorDef::Fast;
  double terrain_alpha = 1.0;
  std::vector<Object> objects;
  std::optional<atmrt_atmosphere_t> atmosphere; // None = AtmosphereDef::us_76(); the function table and spline points it points at
                                                // are the caller's and must outlive generate() (any number of either)
  uint16_t col_begin = 0, col_end = 0;          // pixel-column shard, 0/0 = whole image (leave 0/0 on a multi-device Terrain)

  atmrt_params_t pod(GeneratorDef gen) const {
    atmrt_params_t p{};
    p.position = atmrt_position_t{position.latitude, position.longitude, (int32_t)position.altitude.kind, 0, position.altitude.value};
    p.frame = atmrt_frame_t{frame.direction, frame.tilt, frame.fov, frame.max_distance};
    p.earth = model.pod;
    p.wavelength = wavelength;
    p.simulation_step = simulation_step;
    p.terrain_alpha = terrain_alpha;
    p.straight_rays = straight_rays ? 1 : 0;
    p.generator = (int32_t)gen;
    p.width = width;
    p.height = height;
    p.col_begin = col_begin;
    p.col_end = col_end;
    return p;
  }
};

// Terrain, terrain/mod.rs:55-57.  Owns the device context the tiles live in — one GPU, or several: with a device list the
// library cuts every frame into pixel-column tiles (one per device, one host thread each, the mosaic in every device's HBM) and
// generate() still returns the whole Vec<Vec<ResultPixel>>; nothing else in the host code changes.
class Terrain {
 public:
  explicit Terrain(int device = 0) {
    atmrt_ctx* c = nullptr;
    int rc = atmrt_ctx_create(&c, device);
    if (rc) throw Error(rc, atmrt_last_error(nullptr));
    ctx_.reset(c, atmrt_ctx_destroy);
  }
  explicit Terrain(const std::vector<int>& devices) {
    atmrt_ctx* c = nullptr;
    std::vector<int32_t> d(devices.begin(), devices.end());
    int rc = atmrt_ctx_create_multi(&c, d.data(), (int32_t)d.size());
    if (rc) throw Error(rc, atmrt_last_error(nullptr));
    ctx_.reset(c, atmrt_ctx_destroy);
  }
  static Terrain from_folder(const std::string& terrain_folder, int device = 0) { return Terrain(device).load(terrain_folder); }
  static Terrain from_folder(const std::string& terrain_folder, const std::vector<int>& devices) { return Terrain(devices).load(terrain_folder); }
  int devices() const { return atmrt_ctx_device_count(ctx()); }
  void add_tile(int lat0, int lon0, int n_lat, int n_lon, const int16_t* posts) { check(atmrt_terrain_add_tile(ctx(), lat0, lon0, n_lat, n_lon, posts)); }
  std::optional<double> get_elev(double latitude, double longitude) const {
    double e = 0.0;
    uint8_t ok = 0;
    check(atmrt_terrain_get_elev(ctx(), 1, &latitude, &longitude, &e, &ok));
    return ok ? std::optional<double>(e) : std::nullopt;
  }
  int files() const { return files_; }
  atmrt_ctx* ctx() const { return ctx_.get(); }
  Terrain& load(const std::string& terrain_folder) {
    int32_t n = 0;
    check(atmrt_terrain_load_dir(ctx(), terrain_folder.c_str(), &n));
    files_ = n;
    return *this;
  }
  void check(int rc) const { if (rc) throw Error(rc, atmrt_last_error(ctx())); }
 private:
  std::shared_ptr<atmrt_ctx> ctx_;
  int files_ = 0;
};

struct PixelColor { // generators/mod.rs:45-49
  bool terrain = true;
  Color rgba;        // Terrain(alpha): rgba.a = alpha
  double alpha() const { return rgba.a; }
};
struct TracePoint { // generators/mod.rs:21-30
  double lat, lon, distance, elevation, path_length;
  std::array<double, 3> normal;
  PixelColor color;
};
struct ResultPixel { // generators/mod.rs:13-19
  double elevation_angle, azimuth;
  std::vector<TracePoint> trace_points;
};

struct Generator { // generators/mod.rs:82-84
  virtual ~Generator() = default;
  virtual std::vector<std::vector<ResultPixel>> generate() const = 0;
  uint64_t last_ray_steps = 0;
};

class HipGenerator : public Generator {
 public:
  HipGenerator(const Params& params, const Terrain& terrain, GeneratorDef kind) : params_(params), terrain_(terrain), kind_(kind) {}
  std::vector<std::vector<ResultPixel>> generate() const override {
    atmrt_params_t pod = params_.pod(kind_);
    terrain_.check(atmrt_set_params(terrain_.ctx(), &pod));
    atmrt_atmosphere_t atm;
    if (params_.atmosphere) atm = *params_.atmosphere; else atmrt_atmosphere_us76(&atm);
    terrain_.check(atmrt_set_atmosphere(terrain_.ctx(), &atm));
    std::vector<atmrt_object_t> objs;
    for (const Object& o : params_.objects) {
      atmrt_object_t p = o.pod;
      p.texture_rgba = o.texture.empty() ? nullptr : o.texture.data();
      objs.push_back(p);
    }
    terrain_.check(atmrt_objects_set(terrain_.ctx(), objs.data(), objs.size()));
    atmrt_result_t r{};
    terrain_.check(atmrt_generate(terrain_.ctx(), &r));
    std::vector<std::vector<ResultPixel>> out(r.height);
    for (uint32_t y = 0; y < r.height; y++) {
      out[y].resize(r.width);
      for (uint32_t x = 0; x < r.width; x++) {
        size_t p = (size_t)y * r.width + x;
        ResultPixel& px = out[y][x];
        px.elevation_angle = r.elevation_angle[p];
        px.azimuth = r.azimuth[p];
        for (uint64_t k = r.hit_offset[p]; k < r.hit_offset[p] + r.hit_count[p]; k++) {
          TracePoint tp{r.lat[k], r.lon[k], r.distance[k], r.elevation[k], r.path_length[k],
                        {r.normal[3 * k], r.normal[3 * k + 1], r.normal[3 * k + 2]}, {}};
          tp.color.terrain = r.color_tag[k] == ATMRT_COLOR_TERRAIN;
          tp.color.rgba = Color{r.rgba[4 * k], r.rgba[4 * k + 1], r.rgba[4 * k + 2], r.rgba[4 * k + 3]};
          px.trace_points.push_back(tp);
        }
      }
    }
    const_cast<HipGenerator*>(this)->last_ray_steps = r.ray_steps;
    atmrt_result_free(&r);
    return out;
  }
 private:
  const Params& params_;
  const Terrain& terrain_;
  GeneratorDef kind_;
};

struct FastGenerator : HipGenerator { // fast.rs:102-108
  FastGenerator(const Params& p, const Terrain& t) : HipGenerator(p, t, GeneratorDef::Fast) {}
};
struct RectilinearGenerator : HipGenerator { // rectilinear.rs:70-76
  RectilinearGenerator(const Params& p, const Terrain& t) : HipGenerator(p, t, GeneratorDef::Rectilinear) {}
};
struct InterpolatingRectilinearGenerator : HipGenerator { // interpolating_rectilinear.rs:421-427
  InterpolatingRectilinearGenerator(const Params& p, const Terrain& t) : HipGenerator(p, t, GeneratorDef::InterpolatingRectilinear) {}
};

// generator::generate's `match params.output.generator` (src/generator/mod.rs:72-78)
inline std::unique_ptr<Generator> make_generator(const Params& params, const Terrain& terrain) {
  switch (params.generator) {
    case GeneratorDef::Fast: return std::make_unique<FastGenerator>(params, terrain);
    case GeneratorDef::InterpolatingRectilinear: return std::make_unique<InterpolatingRectilinearGenerator>(params, terrain);
    default: return std::make_unique<RectilinearGenerator>(params, terrain);
  }
}

} // namespace atmrt_host

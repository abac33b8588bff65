// gen_host.cpp — the reference's `generator::generate` flow (src/generator/mod.rs:47-99, minus renderer and metadata)
// written against the C++ host mirror: Terrain::from_folder -> Params -> make_generator -> generate().
// Usage: gen_host TERRAIN_DIR GENERATOR(Fast|Rectilinear|InterpolatingRectilinear) WIDTH HEIGHT OUT.bin [DEVICES]
// DEVICES: comma-separated HIP device ordinals, e.g. 0,1,2,3,4,5,6,7 — the frame is then cut into pixel-column tiles inside the
// library (a device may be listed twice: "0,0" is two tiles on one GPU); the host code below is the same either way.
// Writes per pixel: azimuth, elevation_angle, n_trace_points, then the first trace point (lat lon distance elevation) or
// four NaNs, as float64 — tests/test_host_cpp.py compares the file with the oracle.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "atmrt_host.hpp"

using namespace atmrt_host;

int main(int argc, char** argv) {
  if (argc != 6 && argc != 7) {
    fprintf(stderr, "usage: %s TERRAIN_DIR GENERATOR WIDTH HEIGHT OUT.bin [DEVICES]\n", argv[0]);
    return 2;
  }
  try {
    std::vector<int> devices;
    if (argc == 7)
      for (const char* p = argv[6]; *p;) {
        devices.push_back((int)strtol(p, const_cast<char**>(&p), 10));
        if (*p == ',') p++;
      }
    Terrain terrain = devices.empty() ? Terrain::from_folder(argv[1]) : Terrain::from_folder(argv[1], devices);
    if (!devices.empty()) printf("%d devices\n", terrain.devices());
    printf("Detected %d terrain files\n", terrain.files()); // terrain/mod.rs:80
    Params params;
    params.position = Position{46.5, 8.5, Altitude{Altitude::Relative, 50.0}};
    params.frame = Frame{0.0, -2.0, 60.0, 60000.0};
    params.model = EarthModel::Spherical(6371000.0);
    params.simulation_step = 100.0;
    params.width = (uint16_t)atoi(argv[3]);
    params.height = (uint16_t)atoi(argv[4]);
    params.generator = !strcmp(argv[2], "Fast") ? GeneratorDef::Fast
                       : !strcmp(argv[2], "Rectilinear") ? GeneratorDef::Rectilinear : GeneratorDef::InterpolatingRectilinear;
    auto generator = make_generator(params, terrain);
    auto result = generator->generate();
    FILE* f = fopen(argv[5], "wb");
    if (!f) return 3;
    size_t hits = 0;
    for (const auto& row : result)
      for (const ResultPixel& px : row) {
        double rec[7] = {px.azimuth, px.elevation_angle, (double)px.trace_points.size(), NAN, NAN, NAN, NAN};
        if (!px.trace_points.empty()) {
          const TracePoint& tp = px.trace_points[0];
          rec[3] = tp.lat; rec[4] = tp.lon; rec[5] = tp.distance; rec[6] = tp.elevation;
          hits++;
        }
        fwrite(rec, sizeof rec, 1, f);
      }
    fclose(f);
    printf("%zux%zu pixels, %zu with a trace point, %llu ray-steps\n", result[0].size(), result.size(), hits,
           (unsigned long long)generator->last_ray_steps);
    if (auto e = terrain.get_elev(46.5, 8.5)) printf("elevation under the observer: %.3f m\n", *e);
  } catch (const Error& e) {
    fprintf(stderr, "ERROR: %s\n", e.what()); // main.rs:36-38
    return 1;
  }
  return 0;
}

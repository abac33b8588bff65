// Test-only: exposes the product's host-side GeoTIFF reader (atm-raytracer_amd/csrc/atmrt_tiff.h) to tests/test_geotiff.py.
#include "../../atm-raytracer_amd/csrc/atmrt_tiff.h"

extern "C" int t_tiff_read(const char* path, int want, int16_t* out, char* err, int err_cap) {
  std::vector<int16_t> posts;
  std::string why;
  bool ok = atmrt_tiff::read_dem(path, want, posts, why);
  if (!ok) {
    snprintf(err, (size_t)err_cap, "%s", why.c_str());
    return 0;
  }
  memcpy(out, posts.data(), posts.size() * sizeof(int16_t));
  return 1;
}
extern "C" int t_tiff_coords(const char* name, int* lat, int* lon) { return atmrt_tiff::coords_from_name(name, *lat, *lon) ? 1 : 0; }

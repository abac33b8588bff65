// Test-only: the PRODUCT's object code (atm-raytracer_amd/csrc/atmrt_objects.h — what step_object_impl, object_step_impl and
// close_mask_impl call on the device) compiled for the host and run under sanitizers over a randomised workload that follows the
// kernels' call pattern: object_derive per object, latlon_trig + object_is_close per sample, object_out-of-band filter and
// object_collision per segment, results pushed into a bounded, stably sorted per-step list like step_push keeps.
//   g++     -fsanitize=address,undefined -fno-sanitize-recover=all   (out-of-bounds, signed overflow, float-to-int out of range, ...)
//   clang++ -fsanitize=memory                                         (a read of uninitialised memory that decides a branch or an index)
// "run step_object_impl and close_mask_impl through the CPU build with -fsanitize=undefined and an uninitialised-read
// checker" — the tracer's failure under register pressure (profiles/r04/ipra/README.md) is not undefined behaviour in this code.
// Prints a checksum so the work cannot be optimised away; exit code 0 = no report.
#include "../../atm-raytracer_amd/csrc/atmrt_objects.h"
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
using namespace atmrt;

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() {
  rng_state ^= rng_state << 13;
  rng_state ^= rng_state >> 7;
  rng_state ^= rng_state << 17;
  return rng_state;
}
static double uni(double lo, double hi) { return lo + (hi - lo) * (double)(rnd() >> 11) / 9007199254740992.0; }

constexpr int STEP_CAP = 12; // STEP_CANDIDATES of atmrt_device.h
struct Step {
  int n;
  int kind[STEP_CAP];
  Collision col[STEP_CAP];
};
static void push(Step& sh, double prop, int kind, const Collision* c) { // the insertion step_push does
  if (sh.n >= STEP_CAP) {
    sh.n++;
    return;
  }
  int j = sh.n;
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
  } else {
    sh.col[j].normal = v3(0.0, 0.0, 0.0);
    for (int q = 0; q < 4; q++) sh.col[j].color[q] = 0.0;
  }
  sh.n++;
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? atoi(argv[1]) : 40;
  const int earth_kinds[] = {ATMRT_EARTH_SIMPLE_SPHERE, ATMRT_EARTH_WGS84, ATMRT_EARTH_FLAT_DISTORTED, ATMRT_EARTH_AZIMUTHAL_EQUIDISTANT};
  double checksum = 0.0;
  unsigned long long collisions = 0, closes = 0;
  for (int round = 0; round < rounds; round++) {
    atmrt_earth_model_t m{};
    m.kind = earth_kinds[round % 4];
    Earth e;
    if (earth_resolve(m, e)) return 2;
    const double sim_step = round % 3 == 0 ? 25.0 : 100.0;
    // textures: 2x2 up to 64x64, random bytes incl. fully transparent and fully opaque texels
    std::vector<uint8_t> pool;
    std::vector<ObjectDev> objs(48);
    for (size_t i = 0; i < objs.size(); i++) {
      ObjectDev& o = objs[i];
      o = ObjectDev{};
      o.kind = i % 3 == 2 ? ATMRT_OBJ_BILLBOARD : ATMRT_OBJ_FRUSTUM;
      o.lat = uni(46.40, 46.60);
      o.lon = uni(8.40, 8.60);
      o.elev = uni(-50.0, 3000.0);
      o.height = uni(1.0, 600.0);
      o.width = uni(1.0, 400.0);
      o.r1 = uni(0.5, 150.0);
      o.r2 = i % 5 == 0 ? 0.0 : i % 5 == 1 ? o.r1 : uni(0.0, 150.0); // cone, cylinder, frustum
      for (int q = 0; q < 4; q++) o.color[q] = q == 3 ? (i % 4 == 0 ? 1.0 : i % 4 == 1 ? 0.0 : 0.5) : uni(0.0, 1.0);
      if (o.kind == ATMRT_OBJ_BILLBOARD) {
        o.tex_w = 2 + (int)(rnd() % 63);
        o.tex_h = 2 + (int)(rnd() % 63);
        o.tex_offset = (int64_t)pool.size();
        for (int t = 0; t < o.tex_w * o.tex_h * 4; t++) pool.push_back(t % 4 == 3 ? (uint8_t)((rnd() % 3) * 127 + (rnd() % 2)) : (uint8_t)rnd());
      }
      object_derive(e, sim_step, o);
    }
    // segments: chords between two samples near the objects (so that the proximity filter passes often), any direction incl. vertical
    for (int s = 0; s < 20000; s++) {
      const ObjectDev& near = objs[rnd() % objs.size()];
      const double reach = dm_sqrt(near.close2);
      const double lat0 = near.lat + uni(-1.0, 1.0) * reach / 111000.0, lon0 = near.lon + uni(-1.0, 1.0) * reach / 76000.0;
      const double lat1 = lat0 + uni(-1.0, 1.0) * sim_step / 111000.0, lon1 = lon0 + uni(-1.0, 1.0) * sim_step / 76000.0;
      const double re0 = near.elev + uni(-100.0, near.height + 100.0), re1 = re0 + uni(-60.0, 60.0);
      const LatLonTrig t0 = latlon_trig(e, lat0, lon0), t1 = latlon_trig(e, lat1, lon1);
      const Vec3 pos1 = as_cartesian(e, lat0, lon0, re0), pos2 = as_cartesian(e, lat1, lon1, re1);
      Step sh;
      sh.n = 0;
      if (s % 7 == 0) push(sh, uni(0.0, 1.0), -1, nullptr); // a terrain crossing in the same step
      for (size_t j = 0; j < objs.size(); j++) {
        const ObjectDev& o = objs[j];
        if ((re0 < o.vlo && re1 < o.vlo) || (re0 > o.vhi && re1 > o.vhi)) continue; // object_out_of_band
        if (!(object_is_close(e, o, t0) || object_is_close(e, o, t1))) continue;
        closes++;
        Collision col[4];
        const int nc = object_collision(o, pool.data(), pos1, pos2, col);
        if (nc < 0 || nc > 4) return 3;
        for (int q = 0; q < nc; q++) {
          if (col[q].color[3] == 0.0) continue;
          push(sh, col[q].prop, (int)j, &col[q]);
          collisions++;
          if (col[q].color[3] == 1.0) break;
        }
      }
      const int kept = sh.n < STEP_CAP ? sh.n : STEP_CAP;
      for (int q = 0; q < kept; q++) {
        if (q && sh.col[q - 1].prop > sh.col[q].prop) return 4; // sorted
        checksum += sh.col[q].prop + sh.col[q].normal.x + sh.col[q].color[0] + sh.col[q].color[3] + (double)sh.kind[q];
      }
    }
  }
  printf("objects_san: %llu proximity passes, %llu collisions, checksum %.17g\n", closes, collisions, checksum);
  return collisions > 1000 ? 0 : 5; // the workload must actually reach the collision code
}

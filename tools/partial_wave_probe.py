"""Black-box probe for profiles/r04/ipra/README.md part 2: on a tracer built with the failing flags, is the ray-step count wrong exactly
when the list of object rays does not fill its last wavefront (object_rays % 64 != 0)?  Prints, per scene: object_rays, its remainder
modulo 64, GPU ray_steps - oracle ray_steps, and whether every other field is bit-identical.
  ATMRT_LIB=<failing build> ATMRT_MARCH_VARIANT=plain python tools/partial_wave_probe.py"""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
from atm_raytracer_amd import generators, synth
from util import run_gpu, run_oracle, bits, frame_stats
from oracle_binding import Oracle
o = Oracle("det")
ctx = generators.Context(0)
for w, h, ncyl, spread in ((48, 24, 14, 25.0), (48, 24, 14, 1.5), (64, 16, 10, 20.0), (64, 16, 20, 40.0), (64, 32, 6, 10.0), (96, 16, 30, 50.0),
                           (128, 8, 12, 30.0), (128, 8, 40, 55.0), (64, 8, 8, 12.0), (64, 8, 25, 55.0), (32, 32, 9, 8.0), (80, 20, 18, 35.0)):
    cfg, tiles = synth.scene("S2", w, h, generator="Rectilinear", terrain_alpha=0.5, max_distance=20_000.0, tilt=-2.0)
    synth.add_objects(cfg, n_cyl=ncyl, n_bill=0, dist=(1_500.0, 6_000.0), spread_deg=spread, radius=(30.0, 60.0), height=(300.0, 700.0))
    got = run_gpu(ctx, cfg, tiles); st = frame_stats(ctx); want = run_oracle(o, cfg, tiles)
    same = got["n_hits"] == want["n_hits"] and all(np.array_equal(bits(got[k]), bits(want[k])) for k in ("azimuth", "elevation_angle", "lat", "distance", "rgba", "hit_count"))
    print(f"{w}x{h} cyl {ncyl:3d}: object_rays {st['object_rays']:5d} (mod 64 = {st['object_rays'] % 64:2d})  ray_steps GPU - oracle = {int(got['ray_steps']) - int(want['ray_steps']):9d}  other fields {'identical' if same else 'DIFFERENT'}")

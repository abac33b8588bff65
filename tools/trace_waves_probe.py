import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import numpy as np
from atm_raytracer_amd import generators, synth, config
from util import run_gpu, run_oracle, bits
from oracle_binding import Oracle
o = Oracle("det")
ctx = generators.Context(0)
# the probe scenes of tools/ipra_probe.py, each against the oracle bit for bit incl. the ray-step count (profiles/r04/ipra/README.md part 2;
# tests/test_gpu_march_variants.py runs this file with ATMRT_MARCH_VARIANT=plain)
for name, size, kw, okw in (("scene_objects", (48, 24), dict(terrain_alpha=1.0), dict(n_cyl=14, n_bill=6, dist=(1_000.0, 40_000.0), spread_deg=25.0)),
                            ("per_lane_list", (48, 24), dict(terrain_alpha=0.5, max_distance=20_000.0, tilt=-2.0), dict(n_cyl=14, n_bill=0, dist=(1_500.0, 1_650.0), spread_deg=1.5, radius=(30.0, 60.0), height=(300.0, 700.0))),
                            # one wavefront per row, objects on one side only: the rays WITHOUT candidates are the ones whose azimuth a tracer
                            # with its spill stores ahead of the exec restore writes as 0.0 (profiles/r04/ipra/README.md part 3)
                            ("one_sided", (64, 8), dict(terrain_alpha=0.5, max_distance=20_000.0, tilt=-2.0), dict(n_cyl=8, n_bill=0, dist=(1_500.0, 6_000.0), spread_deg=12.0, radius=(30.0, 60.0), height=(300.0, 700.0)))):
    cfg, tiles = synth.scene("S2", size[0], size[1], generator="Rectilinear", **kw)
    synth.add_objects(cfg, **okw)
    got = run_gpu(ctx, cfg, tiles); want = run_oracle(o, cfg, tiles)
    same = got["n_hits"] == want["n_hits"] and got["ray_steps"] == want["ray_steps"] and all(np.array_equal(bits(got[k]), bits(want[k])) for k in ("azimuth", "elevation_angle", "hit_count", "lat", "distance", "rgba"))
    print(name, "n_hits", got["n_hits"], want["n_hits"], "ray_steps", got["ray_steps"], want["ray_steps"], "identical" if same else "DIFFERENT")

import sys, os, numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..")); sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from atm_raytracer_amd import synth, generators
from util import run_gpu
ctx = generators.Context(0)
for w, h, kw in ((24, 384, dict(step=150.0)), (64, 512, dict()), (16, 1024, dict()), (128, 256, dict(tilt=-1.0)), (96, 96, dict(fov=20.0, tilt=-1.0))):
    cfg, tiles = synth.scene("headline", w, h, generator="Rectilinear", terrain_alpha=0.3, **kw)
    r = run_gpu(ctx, cfg, tiles)
    hc = r["hit_count"]
    print(w, h, kw, "hits", r["n_hits"], "max", hc.max(), "hist", np.bincount(hc.ravel())[:12].tolist(), ">4:", int((hc > 4).sum()), "steps", r["ray_steps"])

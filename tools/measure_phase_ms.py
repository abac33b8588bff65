"""Phase times (library HIP events) of the translucent-terrain headline as a whole frame and as one tile of eight:
   python tools/measure_phase_ms.py   (march_ms, pack_ms: the fill passes after the counting march)"""
import sys, json
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from atm_raytracer_amd import generators, synth
W, H = 4096, 2048
cfg, tiles = synth.scene("headline", W, H, generator="Rectilinear", level=2, terrain_alpha=0.5)
ctx = generators.Context(0)
terrain = generators.Terrain.from_tiles(tiles, ctx)
for G, g in ((1, 0), (8, 3)):
    c0, c1 = g * W // G, (g + 1) * W // G
    cfg.params.col_begin, cfg.params.col_end = c0, c1
    _planes, slab_pod = generators.image_planes(H, c1 - c0, torch.device("cuda", 0))
    gen = generators.make_generator(generators.Params(cfg), terrain)
    for _ in range(3):
        s, ms = gen.generate_device(slab_pod)
    tt = gen.last_timings(); t = {k: round(v, 2) for k, v in tt.items() if k.endswith("_ms")}
    print(G, g, round(ms, 2), t)

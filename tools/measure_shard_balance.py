#!/usr/bin/env python3
"""Load balance of the pixel-column tiles of the headline frame: every one of the G shards a G-GPU run would compute is timed on
ONE GPU (one after the other, best of five back-to-back runs).  mean / max of the shard times is the load balance; the sum of the
shard times against the G = 1 time is what splitting costs the march itself (tail of a smaller grid); both bound the strong-scaling
efficiency from above (the all-gather comes on top).   python tools/measure_shard_balance.py [G ...] [alpha=0.5] [objects=1000]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from atm_raytracer_amd import generators, sharding, synth  # noqa: E402

W, H = 4096, 2048
ALPHA = [float(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("alpha=")]  # alpha=0.5: translucent terrain (the counting march)
OBJECTS = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("objects=")]  # objects=1000 alpha=0.5: BASELINE config 5
cfg, tiles = synth.scene("headline", W, H, generator="Rectilinear", level=2, **(dict(terrain_alpha=ALPHA[0]) if ALPHA else {}))
if OBJECTS:
    synth.add_objects(cfg, n_cyl=int(OBJECTS[0] * 0.7), n_bill=OBJECTS[0] - int(OBJECTS[0] * 0.7))
ctx = generators.Context(0)
terrain = generators.Terrain.from_tiles(tiles, ctx)
out = {}
for G in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4, 8]:
    times, steps = [], []
    for g in range(G):
        c0, c1 = sharding.column_shard(W, g, G)
        cfg.params.col_begin, cfg.params.col_end = c0, c1
        slab = sharding.PlaneSlab(H, c1 - c0, torch.device("cuda", 0))
        gen = generators.make_generator(generators.Params(cfg), terrain)
        gen.generate_device(slab.device_planes())
        runs = [gen.generate_device(slab.device_planes()) for _ in range(5)]  # back to back: the clocks stay up as in a running job
        s, ms = min(runs, key=lambda r: r[1])
        times.append(ms)
        steps.append(s)
    out[G] = {"shard_ms": times, "shard_ray_steps": steps, "max_over_mean": max(times) / (sum(times) / G),
              "balance": (sum(times) / G) / max(times), "sum_ms": sum(times)}
    print(G, [round(t, 1) for t in times], "balance", round(out[G]["balance"], 3), "sum", round(sum(times), 1), file=sys.stderr)
print(json.dumps(out))

#!/usr/bin/env python3
"""Load balance of the pixel-column tiles of the headline frame: every one of the G shards a G-GPU run would compute is timed on
ONE GPU (one after the other, best of five back-to-back runs).  mean / max of the shard times is the load balance; the sum of the
shard times against the G = 1 time is what splitting costs the march itself (tail of a smaller grid); both bound the strong-scaling
efficiency from above (the all-gather comes on top).   python tools/measure_shard_balance.py [G ...] [alpha=0.5] [objects=1000] [recut=3]
recut=N: after the equal tiling, re-cut the tiles N times with the library's own rule (atmrt_tiles_rebalance, what every rank of a
multi-GPU run evaluates on the gathered tile times after a frame) from the times just measured, and measure again."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import ctypes as C  # noqa: E402
from atm_raytracer_amd import generators, synth  # noqa: E402

W, H = 4096, 2048
ALPHA = [float(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("alpha=")]  # alpha=0.5: translucent terrain (the counting march)
OBJECTS = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("objects=")]  # objects=1000 alpha=0.5: BASELINE config 5
cfg, tiles = synth.scene("headline", W, H, generator="Rectilinear", level=2, **(dict(terrain_alpha=ALPHA[0]) if ALPHA else {}))
if OBJECTS:
    synth.add_objects(cfg, n_cyl=int(OBJECTS[0] * 0.7), n_bill=OBJECTS[0] - int(OBJECTS[0] * 0.7))
ctx = generators.Context(0)
terrain = generators.Terrain.from_tiles(tiles, ctx)
out = {}
RECUT = [int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("recut=")]


def measure(cols):
    times, steps = [], []
    for c0, c1 in zip(cols, cols[1:]):
        cfg.params.col_begin, cfg.params.col_end = c0, c1
        _, pod = generators.image_planes(H, c1 - c0, torch.device("cuda", 0))
        gen = generators.make_generator(generators.Params(cfg), terrain)
        gen.generate_device(pod)
        runs = [gen.generate_device(pod) for _ in range(5)]  # back to back: the clocks stay up as in a running job
        s, ms = min(runs, key=lambda r: r[1])
        times.append(ms)
        steps.append(s)
    return times, steps


for G in [int(a) for a in sys.argv[1:] if a.isdigit()] or [1, 2, 4, 8]:
    cols = [g * W // G for g in range(G + 1)]
    times, steps = measure(cols)
    out[G] = {"shard_ms": times, "shard_ray_steps": steps, "max_over_mean": max(times) / (sum(times) / G),
              "balance": (sum(times) / G) / max(times), "sum_ms": sum(times), "cols": cols}
    print(G, [round(t, 1) for t in times], "balance", round(out[G]["balance"], 3), "sum", round(sum(times), 1), file=sys.stderr)
    for it in range(RECUT[0] if RECUT and G > 1 else 0):
        nxt = (C.c_int32 * (G + 1))()
        if max(times) * G <= 1.01 * sum(times):
            break  # the library leaves a tiling that is balanced within 1 % alone
        assert ctx.lib.atmrt_tiles_rebalance(W, G, (C.c_int32 * (G + 1))(*cols), (C.c_double * G)(*times), nxt) == 0
        cols = list(nxt)
        times, steps = measure(cols)
        out[G].setdefault("recut", []).append({"cols": cols, "shard_ms": times, "balance": (sum(times) / G) / max(times), "sum_ms": sum(times),
                                                "max_ms": max(times)})
        print(G, "recut", it + 1, [c1 - c0 for c0, c1 in zip(cols, cols[1:])], [round(t, 1) for t in times], "balance",
              round((sum(times) / G) / max(times), 3), "sum", round(sum(times), 1), "max", round(max(times), 2), file=sys.stderr)
print(json.dumps(out))

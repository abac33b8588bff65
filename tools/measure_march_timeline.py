#!/usr/bin/env python3
"""Experiment: the wavefront timeline of ONE column shard's k_rect_march launch (library built with -DATMRT_TIMELINE, selected
through ATMRT_LIB).  Prints, per millisecond, how many wavefronts are resident and how many SIMDs hold at least 1 / 4 of them;
and the long wavefronts that end last.   ATMRT_LIB=.../dev/tl/libatmrt.so python tools/measure_march_timeline.py [G [g]]"""
import ctypes
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from atm_raytracer_amd import _lib, generators, synth  # noqa: E402

W, H = 4096, 2048
G = int(sys.argv[1]) if len(sys.argv) > 1 else 8
g = int(sys.argv[2]) if len(sys.argv) > 2 else 3
cfg, tiles = synth.scene("headline", W, H, generator="Rectilinear", level=2)
ctx = generators.Context(0)
terrain = generators.Terrain.from_tiles(tiles, ctx)
c0, c1 = g * W // G, (g + 1) * W // G
cfg.params.col_begin, cfg.params.col_end = c0, c1
_planes, slab_pod = generators.image_planes(H, c1 - c0, torch.device("cuda", 0))
gen = generators.make_generator(generators.Params(cfg), terrain)
for _ in range(4):
    steps, ms = gen.generate_device(slab_pod)
n_waves = (c1 - c0) * H // 64
buf = np.zeros(3 * 65536, dtype=np.uint64)
lib = _lib.load()
lib.atmrt_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = lib.atmrt_debug_timeline(buf.ctypes.data, buf.size)
assert rc == 0, rc
t = buf.reshape(-1, 3)[:n_waves]
t0 = t[:, 0].astype(np.int64)
t1 = t[:, 1].astype(np.int64)
base = t0.min()
st = (t0 - base) / 100e3  # 100 MHz -> ms
en = (t1 - base) / 100e3
wsteps = (t[:, 2] & np.uint64(0xffffff)).astype(np.int64)
hw = ((t[:, 2] >> np.uint64(24)) & np.uint64(0xffff)).astype(np.int64)
xcc = ((t[:, 2] >> np.uint64(40)) & np.uint64(0xf)).astype(np.int64)
simd = (hw >> 4) & 3
cu = (hw >> 8) & 15
se = (hw >> 13) & 7
simd_key = ((xcc * 8 + se) * 16 + cu) * 4 + simd
print(f"shard {g} of {G}: {n_waves} wavefronts, event time {ms:.2f} ms, span {en.max():.2f} ms, distinct SIMDs {len(np.unique(simd_key))}")
long_ = wsteps >= 64 * 1900
print(f"long wavefronts (>= 1900 steps on every lane): {long_.sum()}; work in wave-steps: long {wsteps[long_].sum():.3g} of {wsteps.sum():.3g}")
edges = np.arange(0.0, en.max() + 1.0, 1.0)
print(" ms  resident  long  SIMDs>=1  SIMDs>=4  SIMDs>=5")
for a in edges:
    mid = a + 0.5
    on = (st <= mid) & (en > mid)
    keys, counts = np.unique(simd_key[on], return_counts=True)
    print(f"{mid:5.1f} {on.sum():7d} {int((on & long_).sum()):6d} {len(keys):8d} {int((counts >= 4).sum()):8d} {int((counts >= 5).sum()):8d}")
order = np.argsort(-en)[:12]
print("last to end: (wave, row, start, end, duration, steps/lane)")
wl = c1 - c0
for w in order:
    print(f"  {w:6d} row {w * 64 // wl:5d}  {st[w]:7.2f} {en[w]:7.2f} {en[w] - st[w]:7.2f} {wsteps[w] / 64:8.1f}")
dur = en - st
print("duration of long wavefronts by start time: quantiles of start", np.quantile(st[long_], [0, 0.5, 0.9, 0.99, 1]).round(2),
      "duration", np.quantile(dur[long_], [0, 0.1, 0.5, 0.9, 1]).round(2))
late = long_ & (st > 1.0)
if late.any():
    print(f"long wavefronts that started after 1 ms: {late.sum()}, start {st[late].min():.2f}..{st[late].max():.2f}, duration "
          f"{dur[late].min():.2f}..{dur[late].max():.2f}, end {en[late].min():.2f}..{en[late].max():.2f}")
print(json.dumps({"G": G, "g": g, "event_ms": ms, "span_ms": float(en.max()), "waves": int(n_waves), "long": int(long_.sum())}))

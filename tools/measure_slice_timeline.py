#!/usr/bin/env python3
"""Experiment: timeline of the time-sliced march of one column shard (library built with -DATMRT_TIMELINE, ATMRT_SLICED=<steps>)."""
import ctypes, json, os, sys
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
lib = _lib.load()
lib.atmrt_debug_slices.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
sl = np.zeros(4 * 262144 + 8, dtype=np.uint64)
for _ in range(4):
    lib.atmrt_debug_slices(sl.ctypes.data, 8)  # resets the count
    steps, ms = gen.generate_device(slab_pod)
n_waves = (c1 - c0) * H // 64
buf = np.zeros(3 * 65536 + 4 * 32768, dtype=np.uint64)
lib = _lib.load()
lib.atmrt_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
assert lib.atmrt_debug_timeline(buf.ctypes.data, buf.size) == 0
first = buf[:3 * 32768].reshape(-1, 3)[:min(n_waves, 32768)]
base = int(first[:, 0].min())
f0 = (first[:, 0].astype(np.int64) - base) / 1e5
print(f"shard {g}/{G}: event {ms:.2f} ms; first-slice kernel: waves {len(first)}, start {f0.min():.2f}..{f0.max():.2f}")
fin = first[:, 2].astype(np.int64)
fin_ms = np.where(fin > 10**9, (fin - base) / 1e5, np.nan)  # groups that ended in the first-slice kernel keep their step count there
wl = c1 - c0
late = np.argsort(-np.nan_to_num(fin_ms))[:40]
print("  groups that finished last: (group, row, finish ms)")
for gidx in late[:40:3]:
    print(f"    {gidx:6d} row {gidx * 64 // wl:5d} {fin_ms[gidx]:7.2f}")
rows = np.arange(len(fin_ms)) * 64 // wl
for lo in range(0, H, 64):
    m = (rows >= lo) & (rows < lo + 64) & np.isfinite(fin_ms)
    if m.any():
        print(f"    rows {lo:4d}-{lo + 63:4d}: groups {m.sum():4d} finish {np.nanmin(fin_ms[m]):6.2f}..{np.nanmax(fin_ms[m]):6.2f}")

assert lib.atmrt_debug_slices(sl.ctypes.data, sl.size) == 0
ns = int(sl[0]); print("slices logged", ns)
r = sl[8:8 + 4 * min(ns, 262144)].reshape(-1, 4)
sg = (r[:, 0] & np.uint64(0xffffffff)).astype(np.int64); si0 = (r[:, 0] >> np.uint64(32)).astype(np.int64)
s0 = (r[:, 1].astype(np.int64) - base) / 1e5; s1 = (r[:, 2].astype(np.int64) - base) / 1e5
sw = (r[:, 3] & np.uint64(0xffffffff)).astype(np.int64)
dur = s1 - s0
sky = (sg * 64 // wl) < 890
full = sky & (si0 < 1900)
print("  sky slices of 128 steps: duration quantiles ms", np.quantile(dur[full], [0, .01, .1, .5, .9, .99, 1]).round(3))
for a in range(0, 36, 3):
    m = full & (s0 >= a) & (s0 < a + 3)
    if m.any(): print(f"    started {a:2d}-{a+3:2d} ms: n {m.sum():6d} median {np.median(dur[m]):.3f} p99 {np.quantile(dur[m], .99):.3f} max {dur[m].max():.3f}")
# history of the group that finished last
gl = int(np.nanargmax(fin_ms))
m = sg == gl
o = np.argsort(s0[m])
print(f"  history of group {gl} (row {gl * 64 // wl}): (i0, start, end, wave)")
for k in o: print(f"    {si0[m][k]:5d} {s0[m][k]:7.2f} {s1[m][k]:7.2f} {sw[m][k]:5d}")

#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point (atmrt_generate: device compute + packing + copy of every result array to
host memory the library allocates), next to the HBM-resident rate bench.py reports.  Headline workload, one GPU.
Writes one JSON object to stdout; profiles/r01/host_boundary.json is a committed run of it."""
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atm_raytracer_amd import _abi, generators, synth  # noqa: E402


def measure(ctx, cfg, tiles, out, suffix=""):
    terrain = generators.Terrain.from_tiles(tiles, ctx)
    for name in ("Rectilinear", "Fast"):
        cfg.params.generator = _abi.GENERATORS[name]
        gen = generators.make_generator(generators.Params(cfg), terrain)
        gen._configure()
        lib, h = ctx.lib, ctx.handle
        times, nbytes, steps, dev_ms = [], 0, 0, 0.0
        for it in range(4):
            res = _abi.Result()
            t0 = time.perf_counter()
            ctx.check(lib.atmrt_generate(h, C.byref(res)))
            dt = time.perf_counter() - t0
            steps, dev_ms = res.ray_steps, res.device_ms
            nbytes = res.n_pixels * (8 + 8 + 4 + 8) + res.n_hits * (5 * 8 + 24 + 4 + 32)
            lib.atmrt_result_free(C.byref(res))
            if it:
                times.append(dt)
        wall = sum(times) / len(times)
        out[name + suffix] = {"wall_ms": wall * 1e3, "device_ms": dev_ms, "result_bytes": int(nbytes), "ray_steps": int(steps),
                     "ray_steps_per_s_pcie_inclusive": steps / wall, "ray_steps_per_s_device": steps / (dev_ms * 1e-3)}
        if suffix and not suffix.endswith("_config5"):
            out[name + suffix]["comm"] = ctx.comm_timings()
    ctx.close()


def main():
    out = {}
    cfg, tiles = synth.scene("headline", level=2)
    measure(generators.Context(0), cfg, tiles, out)
    # the multi-device route of atmrt_generate (csrc/atmrt_multi.hip) with two / eight sub-contexts on the ONE GPU of this box: the
    # tiles share the device, so the device time is that of the whole frame; what the lines show is the cost of the host-side
    # assembly (strided plane copies + the merge of the trace-point lists on the devices' host threads) on top of it —
    # comm.gather_ms: the device-to-host copies (slowest device), comm.assemble_ms: the merge (wall clock)
    measure(generators.Context.multi([0, 0]), cfg, tiles, out, suffix="_two_tiles_one_gpu")
    measure(generators.Context.multi([0] * 8), cfg, tiles, out, suffix="_eight_tiles_one_gpu")
    # BASELINE config 5: the headline with translucent terrain and 1000 objects (several trace points per pixel: the lists are 4x longer)
    cfg5, _ = synth.scene("headline", level=2, terrain_alpha=0.5)
    synth.add_objects(cfg5)
    measure(generators.Context(0), cfg5, tiles, out, suffix="_config5")
    measure(generators.Context.multi([0] * 8), cfg5, tiles, out, suffix="_config5_eight_tiles_one_gpu")
    print(json.dumps(out))


if __name__ == "__main__":
    main()

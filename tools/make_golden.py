#!/usr/bin/env python3
"""Generate tests/golden/*.npz: inputs (scene document + terrain posts) and expected outputs of the hot path.

The reference cannot be run here (Rust, no toolchain) and ships no fixtures, so the vectors come from the
oracle's glibc-libm flavour — the flavour that shares no numerics with the product.  Re-run:  python tools/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from atm_raytracer_amd import synth  # noqa: E402
from oracle_binding import Oracle  # noqa: E402
from util import run_oracle  # noqa: E402

CASES = {
    "c1_fast_flat_zero_straight": dict(scene="S1", w=64, h=32, generator="Fast"),
    "c1_rect_flat_zero_straight": dict(scene="S1", w=32, h=16, generator="Rectilinear"),
    "c1_fast_flat_distorted": dict(scene="S1", w=48, h=24, generator="Fast", earth_shape="FlatDistorted"),
    "c2_fast_refraction_one_tile": dict(scene="S2", w=64, h=32, generator="Fast", max_distance=120_000.0),
    "c2_rect_refraction_one_tile": dict(scene="S2", w=24, h=16, generator="Rectilinear", max_distance=120_000.0),
    "c2_interp_refraction_one_tile": dict(scene="S2", w=24, h=16, generator="InterpolatingRectilinear", max_distance=120_000.0),
    "c5_fast_translucent_terrain": dict(scene="S2", w=32, h=16, generator="Fast", terrain_alpha=0.5, tilt=-4.0, max_distance=120_000.0),
    "wgs84_rect": dict(scene="S2", w=16, h=12, generator="Rectilinear", earth_shape="Wgs84", max_distance=60_000.0),
}
GOLDEN_LEVEL = 301  # posts per side of the synthetic tile stored in the fixture


def build_case(spec):
    spec = dict(spec)
    cfg, tiles = synth.scene(spec.pop("scene"), spec.pop("w"), spec.pop("h"), generator=spec.pop("generator"), level=GOLDEN_LEVEL, **spec)
    return cfg, tiles


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    oracle = Oracle("libm")
    for name, spec in CASES.items():
        cfg, tiles = build_case(spec)
        res = run_oracle(oracle, cfg, tiles)
        arrays = {k: v for k, v in res.items() if isinstance(v, np.ndarray)}
        meta = {"spec": spec, "ray_steps": res["ray_steps"], "n_hits": res["n_hits"], "tile_keys": [list(k) for k in tiles]}
        tile_arrays = {f"tile_{la}_{lo}": p for (la, lo), p in tiles.items()}
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=json.dumps(meta), **arrays, **tile_arrays)
        print(f"{name}: {res['width']}x{res['height']} px, {res['n_hits']} hits, {res['ray_steps']} ray-steps")


if __name__ == "__main__":
    main()

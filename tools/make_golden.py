#!/usr/bin/env python3
"""Generate tests/golden/*.npz: inputs (scene document + terrain posts) and expected outputs of the hot path.

The reference cannot be run here (Rust, no toolchain) and ships no fixtures, so the vectors come from the
oracle's glibc-libm flavour — the flavour that shares no numerics with the product.  Re-run:  python tools/make_golden.py

The committed vectors were written in round 3, when the oracle rounded every operation of the absent crate's formulas separately.
Round 4 fixed that evaluation order anew (fused multiply-adds, the density form of n(h): oracle/atmosphere.c, oracle/stepper.c) and
the vectors were deliberately NOT regenerated: both flavours of today's oracle and the GPU reproduce them to 1e-9 with identical
hit / miss decisions and step counts (tests/test_golden.py), which is the evidence that the new order moves results by rounding
errors only.  Re-running this script rewrites them under the current order.
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
    "c5_fast_objects_translucent": dict(scene="S2", w=40, h=20, generator="Fast", terrain_alpha=0.5, tilt=-2.0, max_distance=30_000.0, objects=True),
    "c5_rect_objects_opaque": dict(scene="S2", w=24, h=12, generator="Rectilinear", tilt=-2.0, max_distance=30_000.0, objects=True),
    "spline_inversion_fast": dict(scene="S2", w=32, h=24, generator="Fast", tilt=-0.2, fov=8.0, max_distance=80_000.0, spline=True),
}
# frusta only: a billboard's colour and alpha are bilinear texel blends truncated to u8 (object/mod.rs:91-117), so a last-bit
# difference between libm and the deterministic functions can move a channel by one level or toggle an alpha == 0 / == 1 test
# (utils.rs:258, 274); billboards are covered by the bit-exact GPU-vs-oracle tests instead
OBJECTS = dict(n_cyl=36, n_bill=0, dist=(300.0, 6_000.0), spread_deg=28.0, radius=(30.0, 120.0), height=(150.0, 600.0),
               bill_w=(150.0, 500.0), bill_h=(150.0, 500.0))
SPLINE_ATMOSPHERE = {"pressure": {"altitude": 0.0, "pressure": 101325.0},
                     "first_temperature_function": {"Spline": {"boundary_condition": "Natural", "points": [
                         [0.0, 283.15], [40.0, 283.6], [80.0, 287.9], [150.0, 289.2], [400.0, 287.4], [1500.0, 280.2]]}},
                     "next_functions": [{"altitude": 1500.0, "function": {"Linear": {"gradient": -0.0065}}},
                                        {"altitude": 11000.0, "function": {"Linear": {"gradient": 0.0}}}]}
RENDER_VIEW = {"coloring": {"Shading": {"water_level": 700.0, "ambient_light": 0.3, "light_zenith_angle": 55.0, "light_dir": 30.0}},
               "fog_distance": 60_000.0}
GOLDEN_LEVEL = 301  # posts per side of the synthetic tile stored in the fixture


def build_case(spec, with_terrain=True):
    """Config (+ synthetic tiles) of a golden case; `objects` / `spline` switch on the S5-style object set and the inversion
    atmosphere; every case also carries the renderer view RENDER_VIEW for its RGB8 image."""
    from atm_raytracer_amd import config
    spec = dict(spec)
    objects, spline = spec.pop("objects", False), spec.pop("spline", False)
    level = GOLDEN_LEVEL if with_terrain else 2
    saved = synth.synth_tiles
    if not with_terrain:
        synth.synth_tiles = lambda *a, **k: {}
    try:
        cfg, tiles = synth.scene(spec.pop("scene"), spec.pop("w"), spec.pop("h"), generator=spec.pop("generator"), level=level, **spec)
    finally:
        synth.synth_tiles = saved
    if objects:
        synth.add_objects(cfg, **OBJECTS)
    if spline:
        cfg.atmosphere = config._atmosphere(SPLINE_ATMOSPHERE)
    cfg.coloring = config._coloring(RENDER_VIEW)
    return cfg, tiles


def main():
    out_dir = os.path.join(ROOT, "tests", "golden")
    os.makedirs(out_dir, exist_ok=True)
    oracle = Oracle("libm")
    for name, spec in CASES.items():
        cfg, tiles = build_case(spec)
        res = run_oracle(oracle, cfg, tiles)
        arrays = {k: v for k, v in res.items() if isinstance(v, np.ndarray)}
        arrays["image_rgb"] = oracle.draw_image(res, oracle.into_coloring(cfg.params, cfg.coloring))
        meta = {"spec": spec, "ray_steps": res["ray_steps"], "n_hits": res["n_hits"], "tile_keys": [list(k) for k in tiles]}
        tile_arrays = {f"tile_{la}_{lo}": p for (la, lo), p in tiles.items()}
        np.savez_compressed(os.path.join(out_dir, name + ".npz"), meta=json.dumps(meta), **arrays, **tile_arrays)
        print(f"{name}: {res['width']}x{res['height']} px, {res['n_hits']} hits, {res['ray_steps']} ray-steps")


if __name__ == "__main__":
    main()

"""Committed golden vectors (tests/golden/*.npz, written by tools/make_golden.py from the oracle's libm flavour):
both oracle flavours must reproduce them on the CPU, and the HIP path must reproduce them on the GPU."""
import glob
import json
import os
import sys

import numpy as np
import pytest

from atm_raytracer_amd import synth
from util import run_gpu, run_oracle

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load(path):
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import make_golden  # the generating script also rebuilds the Config of a case (terrain comes from the fixture)
    cfg, _ = make_golden.build_case(meta["spec"], with_terrain=False)
    tiles = {tuple(k): z[f"tile_{k[0]}_{k[1]}"] for k in meta["tile_keys"]}
    return cfg, tiles, z, meta


def check(res, z, meta, rtol):
    assert res["n_hits"] == meta["n_hits"] and res["ray_steps"] == meta["ray_steps"]
    assert np.array_equal(res["hit_count"], z["hit_count"]) and np.array_equal(res["hit_offset"], z["hit_offset"])
    assert np.array_equal(res["color_tag"], z["color_tag"])
    for k in ("azimuth", "elevation_angle", "lat", "lon", "distance", "elevation", "path_length", "normal", "rgba"):
        np.testing.assert_allclose(res[k], z[k], rtol=rtol, atol=1e-9, err_msg=k)


def test_fixtures_exist():
    assert len(GOLDEN) >= 11


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
@pytest.mark.parametrize("flavour", ["det", "libm"])
def test_oracle_reproduces_golden(path, flavour, oracle_det, oracle_libm):
    cfg, tiles, z, meta = load(path)
    oracle = oracle_det if flavour == "det" else oracle_libm
    res = run_oracle(oracle, cfg, tiles)
    check(res, z, meta, 1e-9)
    img = oracle.draw_image(res, oracle.into_coloring(cfg.params, cfg.coloring)).astype(np.int16)
    assert np.abs(img - z["image_rgb"].astype(np.int16)).max() <= (0 if flavour == "libm" else 1)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_gpu_reproduces_golden(path, gpu_ctx):
    cfg, tiles, z, meta = load(path)
    import ctypes as C
    from atm_raytracer_amd import config, generators
    res = run_gpu(gpu_ctx, cfg, tiles)
    check(res, z, meta, 1e-9)
    img = generators.draw_image(gpu_ctx, generators.into_coloring(gpu_ctx.lib, cfg.params, cfg.coloring), res["width"], res["height"])
    assert np.abs(img.astype(np.int16) - z["image_rgb"].astype(np.int16)).max() <= 1  # det vs libm: at most one 8-bit level
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_atmosphere(gpu_ctx.handle, C.byref(config.us76())))

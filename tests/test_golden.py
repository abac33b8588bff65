"""Committed golden vectors (tests/golden/*.npz, written by tools/make_golden.py from the oracle's libm flavour):
both oracle flavours must reproduce them on the CPU, and the HIP path must reproduce them on the GPU."""
import glob
import json
import os

import numpy as np
import pytest

from atm_raytracer_amd import synth
from util import run_gpu, run_oracle

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def load(path):
    z = np.load(path)
    meta = json.loads(str(z["meta"]))
    spec = dict(meta["spec"])
    import atm_raytracer_amd.synth as s  # rebuild the Config from the stored spec; the terrain comes from the fixture
    orig = s.synth_tiles
    s.synth_tiles = lambda *a, **k: {}
    try:
        cfg, _ = s.scene(spec.pop("scene"), spec.pop("w"), spec.pop("h"), generator=spec.pop("generator"), **spec)
    finally:
        s.synth_tiles = orig
    tiles = {tuple(k): z[f"tile_{k[0]}_{k[1]}"] for k in meta["tile_keys"]}
    return cfg, tiles, z, meta


def check(res, z, meta, rtol):
    assert res["n_hits"] == meta["n_hits"] and res["ray_steps"] == meta["ray_steps"]
    assert np.array_equal(res["hit_count"], z["hit_count"]) and np.array_equal(res["hit_offset"], z["hit_offset"])
    assert np.array_equal(res["color_tag"], z["color_tag"])
    for k in ("azimuth", "elevation_angle", "lat", "lon", "distance", "elevation", "path_length", "normal", "rgba"):
        np.testing.assert_allclose(res[k], z[k], rtol=rtol, atol=1e-9, err_msg=k)


def test_fixtures_exist():
    assert len(GOLDEN) >= 8


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
@pytest.mark.parametrize("flavour", ["det", "libm"])
def test_oracle_reproduces_golden(path, flavour, oracle_det, oracle_libm):
    cfg, tiles, z, meta = load(path)
    check(run_oracle(oracle_det if flavour == "det" else oracle_libm, cfg, tiles), z, meta, 1e-9)


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_gpu_reproduces_golden(path, gpu_ctx):
    cfg, tiles, z, meta = load(path)
    check(run_gpu(gpu_ctx, cfg, tiles), z, meta, 1e-9)

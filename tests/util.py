"""Helpers shared by the parity tests: run a scene through the HIP library (C ABI) or the oracle."""
import numpy as np

from atm_raytracer_amd import generators

FIELDS_PIXEL = ("azimuth", "elevation_angle", "hit_count", "hit_offset")
FIELDS_HIT = ("lat", "lon", "distance", "elevation", "path_length", "normal", "color_tag", "rgba")


def run_gpu(ctx, cfg, tiles):
    ctx.check(ctx.lib.atmrt_terrain_clear(ctx.handle))
    terrain = generators.Terrain.from_tiles(tiles, ctx)
    gen = generators.make_generator(generators.Params(cfg), terrain)
    return gen.generate()


def run_oracle(oracle, cfg, tiles, n_threads=0):
    t = oracle.terrain_new(tiles)
    try:
        return oracle.generate(cfg.params, cfg.atmosphere, t, cfg.objects, n_threads)
    finally:
        oracle.terrain_free(t)


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.uint64) if a.dtype == np.float64 else a


def assert_bitexact(got, want):
    """Every pixel plane and every trace-point field identical to the last bit."""
    assert got["width"] == want["width"] and got["height"] == want["height"]
    assert got["n_hits"] == want["n_hits"], (got["n_hits"], want["n_hits"])
    assert got["ray_steps"] == want["ray_steps"], (got["ray_steps"], want["ray_steps"])
    for k in FIELDS_PIXEL + FIELDS_HIT:
        g, w = bits(got[k]), bits(want[k])
        assert g.shape == w.shape, k
        bad = np.flatnonzero(g.ravel() != w.ravel())
        assert bad.size == 0, f"{k}: {bad.size} of {g.size} differ, first at {bad[:5]}: {got[k].ravel()[bad[:5]]} vs {want[k].ravel()[bad[:5]]}"


def assert_close(got, want, rtol):
    """North-star tolerance: hit/miss and pixel indices exact, fields within rtol relative."""
    assert np.array_equal(got["hit_count"], want["hit_count"]), "hit/miss differs"
    assert np.array_equal(got["hit_offset"], want["hit_offset"])
    for k in ("azimuth", "elevation_angle"):
        np.testing.assert_allclose(got[k], want[k], rtol=rtol, atol=1e-9)
    for k in ("lat", "lon", "distance", "elevation", "path_length"):
        np.testing.assert_allclose(got[k], want[k], rtol=rtol, atol=1e-6, err_msg=k)
    np.testing.assert_allclose(got["normal"], want["normal"], rtol=rtol, atol=1e-7)

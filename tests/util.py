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


def frame_stats(ctx):
    """atmrt_last_stats of the context's last frame."""
    import ctypes as C
    from atm_raytracer_amd import _abi
    t = _abi.FrameStats()
    ctx.check(ctx.lib.atmrt_last_stats(ctx.handle, C.byref(t)))
    return {k: getattr(t, k) for k, _ in _abi.FrameStats._fields_}


def run_oracle(oracle, cfg, tiles, n_threads=0, rows=None):
    t = oracle.terrain_new(tiles)
    try:
        return oracle.generate(cfg.params, cfg.atmosphere, t, cfg.objects, n_threads, rows)
    finally:
        oracle.terrain_free(t)


def bits(a):
    """The bit pattern of every element; NaNs (any payload, either sign: x86 and gfx950 generate different ones) as one value."""
    a = np.ascontiguousarray(a)
    if a.dtype != np.float64:
        return a
    u = a.view(np.uint64).copy()
    u[np.isnan(a)] = 0x7FF8000000000000
    return u


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


def assert_columns_match(full, want, c0, rows=None, x0=0):
    """`want` is the oracle's frame of the column shard [c0, c0 + w) of the frame `full` (the GPU's; its own column 0 is
    frame column x0): every pixel plane and every trace point of those columns must be identical to the last bit, any
    number of trace points per pixel.  rows = (stride, phase) restricts the comparison to the rows the oracle computed."""
    H, w = want["hit_count"].shape
    ys = np.arange(H) if rows is None else np.arange(rows[1], H, rows[0])
    cols = slice(c0 - x0, c0 - x0 + w)
    for k in ("azimuth", "elevation_angle", "hit_count"):
        g, o = bits(full[k][ys, cols]), bits(want[k][ys])
        assert np.array_equal(g, o), (k, c0, int((g != o).sum()))
    cnt = want["hit_count"][ys].astype(np.int64).ravel()
    goff = np.repeat(full["hit_offset"][ys, cols].astype(np.int64).ravel(), cnt)
    woff = np.repeat(want["hit_offset"][ys].astype(np.int64).ravel(), cnt)
    within = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    gi, wi = goff + within, woff + within
    for k in FIELDS_HIT:
        g, o = bits(full[k][gi]), bits(want[k][wi])
        bad = np.flatnonzero((g != o).reshape(len(gi), -1).any(axis=1)) if len(gi) else np.zeros(0, int)
        assert bad.size == 0, f"{k}: {bad.size} of {len(gi)} trace points differ in columns {c0}..{c0 + w - 1}"
    return int(cnt.sum())


def assert_columns_close(full, want, c0, rtol, rows=None, x0=0):
    """The north-star statement against the oracle flavour that shares no numerics with the product (glibc libm, what Rust's
    f64::sin etc. call): hit/miss and the number of trace points of every pixel of the columns IDENTICAL (a flip would be a
    `diff1 * diff2 < 0` decided differently, utils.rs:222), azimuth / elevation angle and every trace-point field within rtol
    relative.  Returns (trace points compared, worst relative difference seen in lat / lon / distance / elevation)."""
    H, w = want["hit_count"].shape
    ys = np.arange(H) if rows is None else np.arange(rows[1], H, rows[0])
    cols = slice(c0 - x0, c0 - x0 + w)
    flips = int((full["hit_count"][ys, cols] != want["hit_count"][ys]).sum())
    assert flips == 0, f"{flips} pixels of columns {c0}..{c0 + w - 1} differ in hit/miss or trace-point count between the GPU and the libm oracle"
    for k in ("azimuth", "elevation_angle"):
        np.testing.assert_allclose(full[k][ys, cols], want[k][ys], rtol=rtol, atol=1e-9)
    cnt = want["hit_count"][ys].astype(np.int64).ravel()
    goff = np.repeat(full["hit_offset"][ys, cols].astype(np.int64).ravel(), cnt)
    woff = np.repeat(want["hit_offset"][ys].astype(np.int64).ravel(), cnt)
    within = np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt)
    gi, wi = goff + within, woff + within
    worst = 0.0
    for k in ("lat", "lon", "distance", "elevation", "path_length"):
        g, o = full[k][gi], want[k][wi]
        np.testing.assert_allclose(g, o, rtol=rtol, atol=1e-6, err_msg=k)
        if k != "path_length" and len(gi):
            worst = max(worst, float(np.max(np.abs(g - o) / np.maximum(np.abs(o), 1.0))))
    np.testing.assert_allclose(full["normal"][gi], want["normal"][wi], rtol=rtol, atol=1e-7)
    assert np.array_equal(full["color_tag"][gi], want["color_tag"][wi])
    return int(cnt.sum()), worst

"""GPU parity: the HIP path, called through the C ABI, against the CPU oracle on the same inputs.

Bar (BASELINE.json north_star): hit/miss and pixel indices bit-exact; lat/lon/elevation/distance
within 1e-4 relative.  Against the deterministic-math oracle the HIP path is in fact bit-identical
in every f64 field, which is what these tests assert; against the libm oracle (the flavour that
shares no numerics with the product) the north-star tolerance is asserted.
"""
import os

import numpy as np
import pytest

from atm_raytracer_amd import synth
from util import bits, assert_bitexact, assert_close, run_gpu, run_oracle

pytestmark = pytest.mark.gpu

RTOL = 1e-4  # north-star tolerance for lat / lon / elevation / distance


def test_sqrt_and_division_are_ieee(gpu_ctx):
    """The bit-exactness argument needs correctly rounded / and sqrt on gfx950: probe through the
    geodesic harness (AzEq uses sqrt, FlDs division) against numpy on the host."""
    from atm_raytracer_amd import generators, _abi
    import ctypes as C
    cfg, _ = synth.scene("S1", 8, 8, earth_shape="AzimuthalEquidistant")
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_params(gpu_ctx.handle, C.byref(cfg.params)))
    rng = np.random.default_rng(7)
    d = rng.uniform(0, 3e6, 200000)
    lat, lon = generators.coords_at_dist(gpu_ctx, 30.0, 0.0, 0.0, d)
    # start (lat 30, lon 0) -> pos = (r0, 0), heading north = towards the pole = -x: r = |r0 - d|
    r0 = (90.0 - 30.0) * (10000000.0 / 90.0)
    px = r0 + (-1.0) * d
    want = 90.0 - np.sqrt(px * px + (0.0 + (-0.0) * d) ** 2) / (10000000.0 / 90.0)
    assert np.array_equal(lat, want)


@pytest.mark.parametrize("generator", ["Fast", "Rectilinear"])
def test_s1_flat_zero_terrain_straight(gpu_ctx, oracle_det, generator):
    """BASELINE config 1 at its stated size: 256x128, straight rays, no terrain files (every lookup -> 0 m)."""
    cfg, tiles = synth.scene("S1", generator=generator)
    assert (cfg.params.width, cfg.params.height) == (256, 128)
    assert_bitexact(run_gpu(gpu_ctx, cfg, tiles), run_oracle(oracle_det, cfg, tiles))


@pytest.mark.parametrize("generator,w,h", [("Fast", 192, 96), ("Rectilinear", 64, 40)])
def test_s2_refraction_one_tile(gpu_ctx, oracle_det, oracle_libm, generator, w, h):
    """BASELINE config 2 (reduced size): spherical Earth + US-76 refraction, one synthetic DTED tile."""
    cfg, tiles = synth.scene("S2", w, h, generator=generator)
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert got["n_hits"] > 0
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))
    assert_close(got, run_oracle(oracle_libm, cfg, tiles), RTOL)


@pytest.mark.parametrize("generator", ["Fast", "Rectilinear", "InterpolatingRectilinear"])
def test_s2_at_its_stated_size(gpu_ctx, oracle_det, oracle_libm, generator):
    """BASELINE config 2 AS STATED: 1024x512, spherical Earth + US-76 refraction, one synthetic 1x1 degree DTED tile, 100 m steps
    to 200 km.  The Fast oracle computes the whole frame; for the per-pixel generators it computes 24 columns spread over the
    frame (column shards of the same frame).  Bit-exact vs the deterministic oracle, identical hit/miss + 1e-4 vs libm."""
    from util import assert_columns_close, assert_columns_match
    cfg, tiles = synth.scene("S2", generator=generator)
    assert (cfg.params.width, cfg.params.height, cfg.params.simulation_step, cfg.params.frame.max_distance) == (1024, 512, 100.0, 200_000.0)
    full = run_gpu(gpu_ctx, cfg, tiles)
    assert full["hit_count"].shape == (512, 1024) and full["n_hits"] > 10_000
    if generator == "Fast":
        assert_bitexact(full, run_oracle(oracle_det, cfg, tiles))
        assert_close(full, run_oracle(oracle_libm, cfg, tiles), RTOL)
        return
    n = 0
    for c0 in range(3, 1020, 85):
        shard = synth.scene("S2", generator=generator)[0]
        shard.params.col_begin, shard.params.col_end = c0, c0 + 2
        n += assert_columns_match(full, run_oracle(oracle_det, shard, tiles), c0)
        assert_columns_close(full, run_oracle(oracle_libm, shard, tiles), c0, RTOL)
    assert n > 500


@pytest.mark.parametrize("earth", ["SimpleSphere", "Wgs84", {"Ellipsoid": {"a": 6378137.0, "b": 6356752.3}},
                                   "AzimuthalEquidistant", "FlatDistorted", {"ObserverAe": {"proj_radius": 6371000.0}},
                                   "SimpleObserverAe"])
@pytest.mark.parametrize("generator", ["Fast", "Rectilinear"])
def test_every_earth_model(gpu_ctx, oracle_det, earth, generator):
    w, h = (96, 48) if generator == "Fast" else (48, 24)
    cfg, tiles = synth.scene("S2", w, h, generator=generator, earth_shape=earth, max_distance=60_000.0)
    assert_bitexact(run_gpu(gpu_ctx, cfg, tiles), run_oracle(oracle_det, cfg, tiles))


@pytest.mark.parametrize("generator", ["Fast", "Rectilinear"])
def test_translucent_terrain_multi_hit(gpu_ctx, oracle_det, generator):
    """terrain_alpha < 1: every sign change is a trace point, the march does not stop (utils.rs:237)."""
    w, h = (96, 48) if generator == "Fast" else (48, 24)
    cfg, tiles = synth.scene("S2", w, h, generator=generator, terrain_alpha=0.5, tilt=-4.0)
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert got["hit_count"].max() > 1
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))


def test_translucent_rectilinear_more_crossings_than_slots(gpu_ctx, oracle_det):
    """Rectilinear, terrain_alpha < 1: the counting march records the first 4 crossings of a pixel in slots and only pixels with
    more are marched again (k_rect_gather_slots + the pixel-list form of k_rect_march).  A narrow grazing view of the headline
    terrain has both kinds of pixel (up to ~10 crossings), also inside one wavefront."""
    cfg, tiles = synth.scene("headline", 72, 72, generator="Rectilinear", terrain_alpha=0.3, fov=20.0, tilt=-1.0)
    got = run_gpu(gpu_ctx, cfg, tiles)
    over = int((got["hit_count"] > 4).sum())
    assert over > 0 and int((got["hit_count"] > 0).sum()) > over, (over, int(got["hit_count"].max()))
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))
    # a frame whose crossings all fit the slots right after one that overflowed (the overflow list must not leak)
    cfg2, tiles2 = synth.scene("S2", 48, 24, generator="Rectilinear", terrain_alpha=0.5, tilt=-4.0)
    assert_bitexact(run_gpu(gpu_ctx, cfg2, tiles2), run_oracle(oracle_det, cfg2, tiles2))


def test_three_by_three_tiles_and_shards(gpu_ctx, oracle_det):
    """BASELINE config 3 layout (3x3 tiles, 120 deg fov) at reduced size, computed as two column shards."""
    cfg, tiles = synth.scene("S3", 128, 64, step=200.0)
    want = run_oracle(oracle_det, cfg, tiles)
    assert_bitexact(run_gpu(gpu_ctx, cfg, tiles), want)
    halves = []
    for c0, c1 in ((0, 48), (48, 128)):
        cfg.params.col_begin, cfg.params.col_end = c0, c1
        halves.append(run_gpu(gpu_ctx, cfg, tiles))
        assert_bitexact(halves[-1], run_oracle(oracle_det, cfg, tiles))
    az = np.concatenate([r["azimuth"] for r in halves], axis=1)
    assert np.array_equal(az, want["azimuth"])
    assert np.array_equal(np.concatenate([r["hit_count"] for r in halves], axis=1), want["hit_count"])


def test_ray_paths_harness(gpu_ctx, oracle_det):
    """output-ray-paths (ray_path.rs:65-103): the integrator alone, GPU vs oracle, bit-exact."""
    import ctypes as C
    from atm_raytracer_amd import generators
    cfg, _ = synth.scene("S2", 8, 8)
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_params(gpu_ctx.handle, C.byref(cfg.params)))
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_atmosphere(gpu_ctx.handle, C.byref(cfg.atmosphere)))
    ang = np.arange(-1.0, 1.0001, 0.1)
    for straight in (False, True):
        x, h = generators.ray_paths(gpu_ctx, 2.0, ang, 50.0, 400, straight)
        xo, ho = oracle_det.ray_paths(cfg.params, 2.0, ang, 50.0, 400, straight)
        assert np.array_equal(x, xo) and np.array_equal(h, ho)


def test_terrain_get_elev_edges(gpu_ctx, oracle_det):
    """Terrain::get_elev: exact at posts, None off-tile, max-edge inclusive (geotiff.rs:77-85 model)."""
    from atm_raytracer_amd import generators
    tiles = synth.synth_tiles([46], [8])
    gpu_ctx.check(gpu_ctx.lib.atmrt_terrain_clear(gpu_ctx.handle))
    terrain = generators.Terrain.from_tiles(tiles, gpu_ctx)
    rng = np.random.default_rng(3)
    lat = np.concatenate([rng.uniform(45.9, 47.1, 5000), [46.0, 47.0, 46.5, 46.999999999999, np.nan]])
    lon = np.concatenate([rng.uniform(7.9, 9.1, 5000), [8.0, 9.0, 8.5, 8.999999999999, 8.5]])
    elev, valid = terrain.get_elev(lat, lon)
    t = oracle_det.terrain_new(tiles)
    for i in range(lat.size):
        e = oracle_det.get_elev(t, float(lat[i]), float(lon[i]))
        assert valid[i] == (e is not None), (lat[i], lon[i])
        if e is not None:
            assert elev[i] == e or (np.isnan(e) and np.isnan(elev[i]))
    oracle_det.terrain_free(t)


def test_full_size_properties(gpu_ctx):
    """BASELINE headline size (4096x2048, 3x3 tiles, 100 m / 200 km): size-independent properties —
    hits lie inside max_distance, rows are monotone in elevation angle, every hit's interpolated terrain
    elevation is within the tile range, and two runs are bit-identical (no atomics on the data path)."""
    cfg, tiles = synth.scene("headline")
    a = run_gpu(gpu_ctx, cfg, tiles)
    assert a["hit_count"].shape == (2048, 4096) and a["hit_count"].max() == 1
    assert (a["distance"] >= 0).all() and (a["distance"] <= 200_000.0).all()
    assert (a["elevation"] >= 0).all() and (a["elevation"] <= 4000.0).all()
    assert np.all(np.diff(a["elevation_angle"][:, 0]) < 0)
    nrm = np.linalg.norm(a["normal"], axis=1)
    assert np.all(np.abs(nrm - 1.0) < 1e-3)  # interpolated unit normals are not renormalised (utils.rs:117-118)
    b = run_gpu(gpu_ctx, cfg, tiles)
    for k in ("hit_count", "lat", "lon", "distance", "elevation"):
        assert np.array_equal(a[k], b[k])
    assert a["ray_steps"] == b["ray_steps"] > 0


@pytest.mark.parametrize("generator", ["Rectilinear", "Fast", "InterpolatingRectilinear"])
def test_full_size_column_samples_match_oracle(gpu_ctx, oracle_det, generator):
    """BASELINE headline size, all three generators: the oracle computes three 2-column shards of the SAME 4096x2048 frame
    (left edge, an interior pair, right edge: 12,288 full-length rays) and the GPU's full frame must hold exactly those bits
    in those columns — the full-size run checked against the oracle, not only against itself."""
    cfg, tiles = synth.scene("headline", generator=generator)
    full = run_gpu(gpu_ctx, cfg, tiles)
    assert full["hit_count"].shape == (2048, 4096)
    for c0 in (0, 1777, 4094):
        shard = synth.scene("headline", generator=generator)[0]
        shard.params.col_begin, shard.params.col_end = c0, c0 + 2
        want = run_oracle(oracle_det, shard, tiles)
        assert want["hit_count"].shape == (2048, 2)
        for k in ("azimuth", "elevation_angle", "hit_count"):
            assert np.array_equal(bits(full[k][:, c0:c0 + 2]), bits(want[k])), (k, c0)
        # trace points of the shard's pixels, gathered from the full frame's packed list (opaque terrain: <= 1 per pixel)
        sel = full["hit_count"][:, c0:c0 + 2] > 0
        off = full["hit_offset"][:, c0:c0 + 2][sel].astype(np.int64)
        woff = want["hit_offset"][want["hit_count"] > 0].astype(np.int64)
        assert off.size == woff.size > 0
        for k in ("lat", "lon", "distance", "elevation", "path_length", "normal"):
            assert np.array_equal(bits(full[k][off]), bits(want[k][woff])), (k, c0)


def _object_scene(generator, w, h, alpha=1.0, **kw):
    cfg, tiles = synth.scene("S2", w, h, generator=generator, terrain_alpha=alpha, max_distance=30_000.0, tilt=-2.0, **kw)
    synth.add_objects(cfg, n_cyl=40, n_bill=24, dist=(300.0, 6_000.0), spread_deg=28.0, radius=(30.0, 120.0), height=(150.0, 600.0),
                      bill_w=(150.0, 500.0), bill_h=(150.0, 500.0))
    return cfg, tiles


@pytest.mark.parametrize("generator,w,h", [("Fast", 96, 48), ("Rectilinear", 48, 24)])
@pytest.mark.parametrize("alpha", [1.0, 0.5])
def test_scene_objects(gpu_ctx, oracle_det, generator, w, h, alpha):
    """BASELINE config 5 (reduced): cylinders / cones / frusta / textured billboards, opaque and translucent, with opaque
    and translucent terrain — the full get_single_pixel (utils.rs:201-289) incl. per-step sorting and early termination."""
    cfg, tiles = _object_scene(generator, w, h, alpha)
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert (got["color_tag"] == 1).sum() > 20, "the scene must produce object hits"
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))


def test_scene_objects_flat_earth(gpu_ctx, oracle_det):
    cfg, tiles = _object_scene("Fast", 64, 32, 0.5, earth_shape="FlatDistorted")
    assert_bitexact(run_gpu(gpu_ctx, cfg, tiles), run_oracle(oracle_det, cfg, tiles))


@pytest.mark.parametrize("kw", [{}, {"terrain_alpha": 0.5, "tilt": -4.0}, {"earth_shape": "Wgs84"}, {"fov": 100.0, "tilt": -8.0}],
                         ids=["opaque", "translucent", "wgs84", "wide-fov"])
def test_interpolating_rectilinear(gpu_ctx, oracle_det, kw):
    """InterpolatingRectilinear (interpolating_rectilinear.rs): lattice of Fast-style pixels + the 15-case 4-corner blend."""
    cfg, tiles = synth.scene("S2", 72, 40, generator="InterpolatingRectilinear", max_distance=80_000.0, **kw)
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert got["n_hits"] > 0
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))


def test_interpolating_rectilinear_with_objects_and_shard(gpu_ctx, oracle_det):
    cfg, tiles = _object_scene("InterpolatingRectilinear", 64, 32, 0.5)
    cfg.params.col_begin, cfg.params.col_end = 16, 56
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert (got["color_tag"] == 1).sum() > 10
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))


def test_objects_in_a_line_overflow_the_candidate_list(gpu_ctx, oracle_det):
    """40 objects on (almost) one azimuth: more than the 24 per-ray candidates of the pre-filter, so the affected rays fall
    back to testing every object — results must not change."""
    cfg, tiles = synth.scene("S2", 48, 24, generator="Rectilinear", terrain_alpha=0.5, max_distance=30_000.0, tilt=-2.0)
    synth.add_objects(cfg, n_cyl=40, n_bill=0, dist=(500.0, 25_000.0), spread_deg=0.02, radius=(30.0, 60.0), height=(400.0, 900.0))
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert (got["color_tag"] == 1).sum() > 20
    # translucent objects behind one another: pixels with more trace points than the 4 slots of the counting pass AND pixels
    # within them, i.e. both the slot gather and the listed re-trace of k_rect_trace run
    assert (got["hit_count"] > 4).any() and ((got["hit_count"] > 0) & (got["hit_count"] <= 4)).any(), got["hit_count"].max()
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))


def test_more_close_objects_than_the_per_lane_list(gpu_ctx, oracle_det):
    """14 cylinders inside one 150 m stretch: the samples next to them see more close objects (TerrainData.objects_close,
    utils.rs:74-80) than the 8 the Rectilinear tracer lists per lane; those steps take the unlisted path — same trace points."""
    cfg, tiles = synth.scene("S2", 48, 24, generator="Rectilinear", terrain_alpha=0.5, max_distance=20_000.0, tilt=-2.0)
    synth.add_objects(cfg, n_cyl=14, n_bill=0, dist=(1_500.0, 1_650.0), spread_deg=1.5, radius=(30.0, 60.0), height=(300.0, 700.0))
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert (got["color_tag"] == 1).sum() > 20
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))


@pytest.mark.parametrize("kw", [dict(generator="Rectilinear", tilt=55.0, fov=60.0, max_distance=400_000.0, step=500.0),
                                dict(generator="Fast", tilt=55.0, fov=60.0, max_distance=400_000.0, step=500.0),
                                dict(generator="Rectilinear", tilt=-60.0, fov=50.0, max_distance=50_000.0),
                                dict(generator="Rectilinear", tilt=-2.0, max_distance=900_000.0, step=20_000.0)],
                         ids=["above-the-atmosphere-rect", "above-the-atmosphere-fast", "steep-down", "20-km-steps"])
def test_out_of_envelope_rays(gpu_ctx, oracle_det, kw):
    """Rays that leave the envelope the in-range division / sqrt / exp-log shortcuts are argued for: 300 km up, where the
    linear continuation of US-76 reaches T <= 0 and n(h) is NaN; a view straight down; 20 km steps (chord sagitta of metres).
    The GPU must still agree with the oracle bit for bit, NaN propagation and step counts included."""
    cfg, tiles = synth.scene("S2", 40, 24, **kw)
    assert_bitexact(run_gpu(gpu_ctx, cfg, tiles), run_oracle(oracle_det, cfg, tiles))


EARTHS = ["SimpleSphere", {"Spherical": {"radius": 6371000.0}}, {"Spherical": {"radius": 3.0e6}}, "Wgs84",
          {"Ellipsoid": {"a": 6378137.0, "b": 6300000.0}}, "AzimuthalEquidistant", "FlatDistorted",
          {"ObserverAe": {"proj_radius": 6371000.0}}, "SimpleObserverAe"]


_SEED0 = int(os.environ.get("ATMRT_RANDOM_SEED_FIRST", "0"))


@pytest.mark.parametrize("seed", range(_SEED0, _SEED0 + int(os.environ.get("ATMRT_RANDOM_SEEDS", "120"))))
def test_randomised_configurations(gpu_ctx, oracle_det, seed):
    """Seeded random sweep over the parameter space (earth model, direction incl. the 0/360 wrap, tilt, field of view, observer
    altitude kind, non-integer steps whose accumulated distances round, straight / refracted, opaque / translucent, generator,
    wavelength, and — for a third of the seeds — a random atmosphere: 1-4 Linear layers with lapse, isothermal and inversion
    gradients, or a Spline temperature profile — and for a quarter random scene objects): every f64 field and every hit decision must match the oracle bit for bit.
    ATMRT_RANDOM_SEEDS widens the sweep and ATMRT_RANDOM_SEED_FIRST moves it (round 2: seed 4899 of a 6000-seed sweep found the
    pathological atmosphere of test_pathological_atmosphere_is_still_bit_exact; seeds 0 .. 59999 pass on the final kernels)."""
    rng = np.random.default_rng(1000 + seed)
    gen = ["Fast", "Rectilinear", "InterpolatingRectilinear"][seed % 3]
    w, h = int(rng.integers(3, 70)), int(rng.integers(2, 40))
    if gen == "Rectilinear":
        w, h = min(w, 40), min(h, 24)
    alt_kind = "Relative" if rng.uniform() < 0.5 else "Absolute"
    doc = {
        "view": {"position": {"latitude": float(rng.uniform(46.2, 46.8)), "longitude": float(rng.uniform(8.2, 8.8)),
                              "altitude": {alt_kind: float(rng.uniform(1.0, 900.0) + (2500.0 if alt_kind == "Absolute" else 0.0))}},
                 "frame": {"direction": float(rng.choice([rng.uniform(-30, 390), 0.0, 359.9, 180.0])), "tilt": float(rng.uniform(-25, 6)),
                           "fov": float(rng.choice([rng.uniform(0.5, 140.0), 30.0])), "max_distance": float(rng.uniform(3_000.0, 70_000.0))}},
        "earth_shape": EARTHS[int(rng.integers(len(EARTHS)))],
        "straight_rays": bool(rng.uniform() < 0.25),
        "simulation_step": float(rng.choice([rng.uniform(37.0, 400.0), 50.0, 33.3])),
        "wavelength": float(rng.uniform(400e-9, 700e-9)),
        "scene": {"terrain_alpha": float(rng.choice([1.0, 1.0, 0.5, 0.05]))},
        "output": {"width": w, "height": h, "generator": gen},
    }
    if seed % 3 == 1 or seed >= 24:
        ra = np.random.default_rng(7000 + seed)
        if ra.uniform() < 0.3:
            knots = np.sort(ra.uniform(-500.0, 30_000.0, int(ra.integers(3, 7))))
            knots[0] = -500.0
            temps = 288.0 - 0.0055 * knots + ra.uniform(-6.0, 6.0, knots.size)
            first = {"Spline": {"boundary_condition": "Natural", "points": [[float(a), float(t)] for a, t in zip(knots, temps)]}}
            doc["atmosphere"] = {"pressure": {"altitude": 0.0, "pressure": float(ra.uniform(950.0, 1040.0)) * 100.0},
                                 "first_temperature_function": first}
        else:
            grads = [float(ra.choice([-0.0065, -0.0098, 0.0, 0.003, -0.002, float(ra.uniform(-0.009, 0.004))])) for _ in range(int(ra.integers(1, 5)))]
            alts = np.sort(ra.uniform(300.0, 25_000.0, len(grads) - 1))
            doc["atmosphere"] = {"pressure": {"altitude": float(ra.uniform(0.0, 500.0)), "pressure": float(ra.uniform(950.0, 1040.0)) * 100.0},
                                 "temperature_fixed_point": {"altitude": float(ra.uniform(0.0, 2000.0)), "temperature": float(ra.uniform(255.0, 305.0))},
                                 "first_temperature_function": {"Linear": {"gradient": grads[0]}},
                                 "next_functions": [{"altitude": float(a), "function": {"Linear": {"gradient": g}}} for a, g in zip(alts, grads[1:])]}
    from atm_raytracer_amd import config
    cfg = config.Config.from_dict(doc)
    if seed % 4 == 3 or seed >= 36:  # scene objects in a quarter of the sweep (every generator, opaque and translucent terrain)
        ro = np.random.default_rng(9000 + seed)
        if seed % 4 == 3 or ro.uniform() < 0.4:
            synth.add_objects(cfg, n_cyl=int(ro.integers(3, 40)), n_bill=int(ro.integers(0, 12)), dist=(200.0, float(ro.uniform(2_000.0, 20_000.0))),
                              spread_deg=float(ro.uniform(5.0, 60.0)), radius=(20.0, 150.0), height=(100.0, 700.0), bill_w=(100.0, 500.0),
                              bill_h=(100.0, 500.0), seed=int(ro.integers(1 << 30)))
    tiles = synth.synth_tiles([46], [8], level=301)
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))


EXTREME_EARTHS = EARTHS + [{"Spherical": {"radius": 2.0e5}}, {"Spherical": {"radius": 6.0e8}}, {"Ellipsoid": {"a": 6378137.0, "b": 5.0e6}},
                           {"Ellipsoid": {"a": 3.0e6, "b": 3.3e6}}, {"ObserverAe": {"proj_radius": 1.0e6}}]


_MOSAIC = {}


def _mosaic_2x2():
    if not _MOSAIC:
        _MOSAIC.update(synth.synth_tiles([46, 47], [8, 9], level=301))
    return _MOSAIC


@pytest.mark.parametrize("seed", range(_SEED0, _SEED0 + int(os.environ.get("ATMRT_EXTREME_SEEDS", "60"))))
def test_randomised_extremes(gpu_ctx, oracle_det, seed):
    """A second seeded sweep over the corners the first one leaves out: views straight up and straight down and fields of view up to
    170 degrees, observers below the terrain, far above it and below sea level, rays that leave the tile, two-sample rays
    (max_distance barely above the step) and thousands of short steps, fully transparent terrain (every crossing recorded), splines
    with Derivatives / SecondDerivatives boundary conditions and lapse rates up to +-50 K/km, small and huge planets and squashed
    ellipsoids, column shards (col_begin / col_end), objects above, below and around the observer incl. the observer INSIDE a
    cylinder, translucent and opaque, and 2 x 2 tile mosaics with a missing tile seen from their common corner.  GPU == oracle in
    every bit; then the frame is drawn (k_draw_image) with a random colouring, palette, light, water level and fog: == the
    oracle's image in every byte."""
    from atm_raytracer_amd import config, _abi
    rng = np.random.default_rng(500_000 + seed)
    gen = ["Fast", "Rectilinear", "InterpolatingRectilinear"][seed % 3]
    w, h = int(rng.integers(2, 48)), int(rng.integers(2, 32))
    if gen == "Rectilinear":
        w, h = min(w, 32), min(h, 20)
    alt_kind = str(rng.choice(["Relative", "Absolute", "Absolute"]))
    alt = float(rng.choice([rng.uniform(0.5, 50.0), rng.uniform(50.0, 9000.0), rng.uniform(-400.0, 0.0), rng.uniform(9000.0, 60000.0)]))
    step = float(rng.choice([rng.uniform(5.0, 40.0), rng.uniform(40.0, 1500.0), 100.0, 7.3]))
    n_steps = float(rng.choice([1.2, 2.5, rng.uniform(3.0, 60.0), rng.uniform(60.0, 900.0)]))
    doc = {
        "view": {"position": {"latitude": float(rng.choice([rng.uniform(46.01, 46.99), 46.0005, 46.9995])),
                              "longitude": float(rng.choice([rng.uniform(8.01, 8.99), 8.0005, 8.9995])), "altitude": {alt_kind: alt}},
                 "frame": {"direction": float(rng.uniform(-400.0, 800.0)), "tilt": float(rng.choice([rng.uniform(-89.0, 89.0), -90.0, 90.0, 0.0])),
                           "fov": float(rng.choice([rng.uniform(0.01, 20.0), rng.uniform(20.0, 170.0)])), "max_distance": step * n_steps}},
        "earth_shape": EXTREME_EARTHS[int(rng.integers(len(EXTREME_EARTHS)))],
        "straight_rays": bool(rng.uniform() < 0.3),
        "simulation_step": step,
        "wavelength": float(rng.choice([rng.uniform(300e-9, 1100e-9), 530e-9])),
        "scene": {"terrain_alpha": float(rng.choice([1.0, 0.5, 0.0, 0.999]))},
        "output": {"width": w, "height": h, "generator": gen},
    }
    u = rng.uniform()
    if u < 0.35:
        n_knots = int(rng.integers(2, 9))
        knots = np.sort(rng.uniform(-1000.0, 40_000.0, n_knots))
        temps = 288.0 - 0.006 * knots + rng.uniform(-15.0, 15.0, n_knots)
        bc = rng.choice(["Natural", "Derivatives", "SecondDerivatives"])
        bcv = "Natural" if bc == "Natural" else {str(bc): [float(rng.uniform(-0.01, 0.01)) if bc == "Derivatives" else float(rng.uniform(-1e-6, 1e-6)),
                                                           float(rng.uniform(-0.01, 0.01)) if bc == "Derivatives" else float(rng.uniform(-1e-6, 1e-6))]}
        doc["atmosphere"] = {"pressure": {"altitude": float(rng.uniform(-200.0, 3000.0)), "pressure": float(rng.uniform(300.0, 1100.0)) * 100.0},
                             "first_temperature_function": {"Spline": {"boundary_condition": bcv, "points": [[float(a), float(t)] for a, t in zip(knots, temps)]}}}
    elif u < 0.7:
        grads = [float(rng.choice([-0.0065, 0.0, 0.05, -0.05, -0.0342, float(rng.uniform(-0.02, 0.02)), 1e-9])) for _ in range(int(rng.integers(1, 7)))]
        alts = np.sort(rng.uniform(-500.0, 50_000.0, len(grads) - 1))
        doc["atmosphere"] = {"pressure": {"altitude": float(rng.uniform(-300.0, 5000.0)), "pressure": float(rng.uniform(200.0, 1100.0)) * 100.0},
                             "temperature_fixed_point": {"altitude": float(rng.uniform(-300.0, 12000.0)), "temperature": float(rng.uniform(180.0, 330.0))},
                             "first_temperature_function": {"Linear": {"gradient": grads[0]}},
                             "next_functions": [{"altitude": float(a), "function": {"Linear": {"gradient": g}}} for a, g in zip(alts, grads[1:])]}
    cfg = config.Config.from_dict(doc)
    if rng.uniform() < 0.5:  # a column shard of the frame
        c0 = int(rng.integers(0, w))
        cfg.params.col_begin, cfg.params.col_end = c0, int(rng.integers(c0 + 1, w + 1))
    if rng.uniform() < 0.45:
        reach = step * n_steps
        synth.add_objects(cfg, n_cyl=int(rng.integers(1, 30)), n_bill=int(rng.integers(0, 10)), dist=(0.0, float(max(reach, 30.0))),
                          spread_deg=float(rng.uniform(1.0, 180.0)), radius=(1.0, float(rng.uniform(5.0, 400.0))), height=(1.0, float(rng.uniform(5.0, 3000.0))),
                          bill_w=(1.0, 600.0), bill_h=(1.0, 600.0), seed=int(rng.integers(1 << 30)))
        for o in cfg.objects:  # some objects float or are sunk, some altitudes are absolute
            if rng.uniform() < 0.3:
                o.position.altitude = float(rng.uniform(-300.0, 2000.0))
            if rng.uniform() < 0.2:
                o.position.altitude_kind = _abi.ALT_ABSOLUTE
                o.position.altitude = float(rng.uniform(0.0, 5000.0))
    if rng.uniform() < 0.3:  # a 2 x 2 mosaic with a hole, the observer near the common corner: rays cross tile borders and the hole
        tiles = dict(_mosaic_2x2())
        del tiles[sorted(tiles)[int(rng.integers(4))]]
        cfg.params.position.latitude = float(47.0 + rng.uniform(-0.02, 0.02))
        cfg.params.position.longitude = float(9.0 + rng.uniform(-0.02, 0.02))
    else:
        tiles = synth.synth_tiles([46], [8], level=301)
    want = run_oracle(oracle_det, cfg, tiles)
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert_bitexact(got, want)
    # and the image of that frame (renderer::draw_image, renderer/mod.rs:367-414) under a random colouring, byte for byte
    from atm_raytracer_amd import generators, _abi as abi
    view = {"fog_distance": float(rng.uniform(1.0, 1.0 + 2.0 * step * n_steps))} if rng.uniform() < 0.5 else {}
    if rng.uniform() < 0.4:
        view["coloring"] = {"Simple": {"water_level": float(rng.uniform(-100.0, 3000.0))}}
    elif rng.uniform() < 0.8:
        view["coloring"] = {"Shading": {"water_level": float(rng.uniform(-100.0, 3000.0)), "ambient_light": float(rng.uniform(0.0, 1.0)),
                                        "light_zenith_angle": float(rng.uniform(0.0, 120.0)), "light_dir": float(rng.uniform(-360.0, 360.0)),
                                        "palette": str(rng.choice(sorted(abi.PALETTES)))}}
    cfg.coloring = config._coloring(view)
    col = generators.into_coloring(gpu_ctx.lib, cfg.params, cfg.coloring)
    ocol = oracle_det.into_coloring(cfg.params, cfg.coloring)
    assert bytes(col) == bytes(ocol)
    assert np.array_equal(generators.draw_image(gpu_ctx, col, got["width"], got["height"]), oracle_det.draw_image(want, ocol))


@pytest.mark.parametrize("seed", range(_SEED0, _SEED0 + int(os.environ.get("ATMRT_LONG_ATM_SEEDS", "30"))))
def test_randomised_long_atmospheres(gpu_ctx, oracle_det, seed):
    """A fourth seeded sweep, for what ABI 4 made possible: atmosphere definitions far beyond the old capacities (8 functions, 32
    knots, 64 segments) — 9 .. 70 temperature functions, Linear ones and Splines of 2 .. 120 knots mixed, thin and thick layers,
    lapse rates of either sign — through all three generators (frames of a few hundred pixels, rays that climb through many
    segments) and through the sampler harness at every segment boundary.  The table is header + records in HBM with a bisection
    layer search; the certificate runs over every segment.  GPU == oracle in every bit."""
    from atm_raytracer_amd import config, generators
    rng = np.random.default_rng(900_000 + seed)
    n_fn = int(rng.integers(9, 71)) if rng.uniform() < 0.7 else int(rng.integers(1, 4))
    tops = np.sort(rng.uniform(0.0, 45_000.0, n_fn - 1)) + np.arange(n_fn - 1) * 0.5
    functions, t_here = [], float(rng.uniform(270.0, 310.0))
    for j in range(n_fn):
        lo = -2000.0 if j == 0 else float(tops[j - 1])
        hi = float(tops[j]) if j < n_fn - 1 else lo + float(rng.uniform(2000.0, 30_000.0))
        if rng.uniform() < (0.25 if n_fn > 3 else 0.9):  # a Spline over (and a little beyond) this function's range
            n_k = int(rng.integers(2, 121 if n_fn <= 3 else 25))
            ks = np.sort(rng.uniform(lo - 50.0, hi + 50.0, n_k)) + np.arange(n_k) * 1e-2
            ts = np.clip(t_here - 0.005 * (ks - lo) + rng.normal(0.0, 1.0, n_k), 150.0, 340.0)
            functions.append({"Spline": {"boundary_condition": "Natural", "points": [[float(a), float(t)] for a, t in zip(ks, ts)]}})
            t_here = float(ts[-1])
        else:
            g = float(rng.choice([-0.0065, 0.0, 0.003, -0.0098, float(rng.uniform(-0.012, 0.012))]))
            functions.append({"Linear": {"gradient": g}})
            t_here = float(np.clip(t_here + g * (hi - max(lo, 0.0)), 160.0, 330.0))
    atm = {"pressure": {"altitude": float(rng.uniform(0.0, 1500.0)), "pressure": float(rng.uniform(700.0, 1050.0)) * 100.0},
           "temperature_fixed_point": {"altitude": float(rng.uniform(0.0, 3000.0)), "temperature": float(rng.uniform(250.0, 300.0))},
           "first_temperature_function": functions[0],
           "next_functions": [{"altitude": float(a), "function": f} for a, f in zip(tops, functions[1:])]}
    gen = ["Fast", "Rectilinear", "InterpolatingRectilinear"][seed % 3]
    w, h = (int(rng.integers(8, 40)), int(rng.integers(6, 24))) if gen != "Rectilinear" else (int(rng.integers(4, 20)), int(rng.integers(4, 14)))
    cfg, tiles = synth.scene("S2", w, h, generator=gen, atmosphere=atm, tilt=float(rng.uniform(-3.0, 25.0)), fov=float(rng.uniform(2.0, 60.0)),
                             max_distance=float(rng.uniform(20_000.0, 150_000.0)), step=float(rng.choice([50.0, 100.0, 400.0])),
                             terrain_alpha=float(rng.choice([1.0, 0.5])))
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))
    env = oracle_det.env(cfg.atmosphere, cfg.params.wavelength)
    edges = np.array([env.from_[k] for k in range(env.n)])
    alt = np.concatenate([edges, np.nextafter(edges, -np.inf), edges + 0.01, rng.uniform(-3000.0, 80_000.0, 200)])
    s = generators.atmosphere_sample(gpu_ctx, alt)
    for i, a in enumerate(alt):
        for key, fn in (("temperature", oracle_det.temperature), ("pressure", oracle_det.pressure), ("n", oracle_det.n), ("dn_dh", oracle_det.dn)):
            x, y = s[key][i], fn(env, a)
            assert x == y or (x != x and y != y), (key, a, x, y)
    import ctypes as C
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_atmosphere(gpu_ctx.handle, C.byref(config.us76())))


@pytest.mark.parametrize("seed", range(_SEED0, _SEED0 + int(os.environ.get("ATMRT_HARNESS_SEEDS", "40"))))
def test_randomised_harnesses(gpu_ctx, oracle_det, seed):
    """The diagnostic entry points (output-atm, output-ray-paths, DirectionalCalc::coords_at_dist: src/atm_printer.rs:37-46,
    ray_path.rs:65-103, directional_calc.rs:5-7) under random atmospheres, earth models, start heights, angles and steps:
    temperature, pressure, n, dn/dh, every point of every path and every geodesic point equal to the oracle's, NaN as NaN."""
    import ctypes as C
    from atm_raytracer_amd import config, generators
    from util import bits
    rng = np.random.default_rng(900_000 + seed)
    doc = {"earth_shape": EXTREME_EARTHS[int(rng.integers(len(EXTREME_EARTHS)))], "simulation_step": 50.0,
           "wavelength": float(rng.uniform(300e-9, 1100e-9)), "output": {"width": 8, "height": 8}}
    u = rng.uniform()
    if u < 0.4:
        n_knots = int(rng.integers(2, 9))
        knots = np.sort(rng.uniform(-1000.0, 40_000.0, n_knots))
        temps = 288.0 - 0.006 * knots + rng.uniform(-15.0, 15.0, n_knots)
        doc["atmosphere"] = {"pressure": {"altitude": float(rng.uniform(-200.0, 3000.0)), "pressure": float(rng.uniform(300.0, 1100.0)) * 100.0},
                             "first_temperature_function": {"Spline": {"boundary_condition": "Natural", "points": [[float(a), float(t)] for a, t in zip(knots, temps)]}}}
    elif u < 0.8:
        grads = [float(rng.choice([-0.0065, 0.0, 0.05, -0.05, -0.0342, float(rng.uniform(-0.02, 0.02)), 1e-9])) for _ in range(int(rng.integers(1, 7)))]
        alts = np.sort(rng.uniform(-500.0, 50_000.0, len(grads) - 1))
        doc["atmosphere"] = {"pressure": {"altitude": float(rng.uniform(-300.0, 5000.0)), "pressure": float(rng.uniform(200.0, 1100.0)) * 100.0},
                             "temperature_fixed_point": {"altitude": float(rng.uniform(-300.0, 12000.0)), "temperature": float(rng.uniform(180.0, 330.0))},
                             "first_temperature_function": {"Linear": {"gradient": grads[0]}},
                             "next_functions": [{"altitude": float(a), "function": {"Linear": {"gradient": g}}} for a, g in zip(alts, grads[1:])]}
    cfg = config.Config.from_dict(doc)
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_params(gpu_ctx.handle, C.byref(cfg.params)))
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_atmosphere(gpu_ctx.handle, C.byref(cfg.atmosphere)))
    try:
        # sampler: sorted runs (whole wavefronts inside one layer: the certified path) and a shuffled tail (per-lane layers)
        alt = np.concatenate([np.sort(rng.uniform(-3000.0, 120_000.0, 1536)), rng.uniform(-3000.0, 400_000.0, 512)])
        got = generators.atmosphere_sample(gpu_ctx, alt)
        env = oracle_det.env(cfg.atmosphere, cfg.params.wavelength)
        for k, f in (("temperature", oracle_det.temperature), ("pressure", oracle_det.pressure), ("n", oracle_det.n), ("dn_dh", oracle_det.dn)):
            want = np.array([f(env, h) for h in alt], dtype=np.float64)
            bad = np.flatnonzero(bits(got[k]) != bits(want))
            assert bad.size == 0, (k, bad.size, alt[bad[:4]], got[k][bad[:4]], want[bad[:4]])
        # integrator
        ang = np.concatenate([rng.uniform(-89.0, 89.0, 6), rng.uniform(-2.0, 2.0, 6), [0.0, 90.0]])
        h0 = float(rng.choice([rng.uniform(0.0, 50.0), rng.uniform(50.0, 12_000.0), rng.uniform(-300.0, 0.0)]))
        step = float(rng.choice([rng.uniform(1.0, 50.0), rng.uniform(50.0, 2000.0), 100.0]))
        straight = bool(rng.uniform() < 0.3)
        n_steps = int(rng.integers(2, 400))
        x, h = generators.ray_paths(gpu_ctx, h0, ang, step, n_steps, straight)
        xo, ho = oracle_det.ray_paths(cfg.params, h0, ang, step, n_steps, straight, cfg.atmosphere)
        assert np.array_equal(bits(x), bits(xo)) and np.array_equal(bits(h), bits(ho))
        # geodesic points
        d = np.concatenate([[0.0], rng.uniform(0.0, 500_000.0, 40), rng.uniform(0.0, 50.0, 8)])
        lat0, lon0, dr = float(rng.uniform(-89.0, 89.0)), float(rng.uniform(-180.0, 180.0)), float(rng.uniform(-400.0, 800.0))
        lat, lon = generators.coords_at_dist(gpu_ctx, lat0, lon0, dr, d)
        want = oracle_det.coords_at_dist(cfg.params.earth, lat0, lon0, dr, d)
        assert np.array_equal(bits(lat), bits(want[:, 0])) and np.array_equal(bits(lon), bits(want[:, 1]))
    finally:
        us = config.us76()
        gpu_ctx.check(gpu_ctx.lib.atmrt_set_atmosphere(gpu_ctx.handle, C.byref(us)))
        d0 = config.Config.from_dict({"output": {"width": 8, "height": 8}})
        gpu_ctx.check(gpu_ctx.lib.atmrt_set_params(gpu_ctx.handle, C.byref(d0.params)))


WILD_SPLINE = {"pressure": {"altitude": 0.0, "pressure": 102390.63927278577},
               "first_temperature_function": {"Spline": {"boundary_condition": "Natural", "points": [
                   [-500.0, 295.22010975963303], [16672.2152178729, 192.93808594359285], [16688.49864235047, 195.36762037322703],
                   [21728.827855811145, 170.01112581350702], [28892.825051123004, 125.19791348512327]]}}}


# seed 31148 of the sweep: the same kind of spline, steeper — the pressure at the base of its upper knot intervals is inf / NaN.
# Until round 2 atmrt_set_atmosphere rejected it ("non-positive temperature or pressure"); the reference has no such check.
OVERFLOWING_SPLINE = {"pressure": {"altitude": 0.0, "pressure": 99730.24971796537},
                      "first_temperature_function": {"Spline": {"boundary_condition": "Natural", "points": [
                          [-500.0, 295.9785296912339], [13886.76872258016, 216.67442822116288], [13916.894125578941, 207.95894584371578],
                          [22713.29131575354, 163.34485706767265], [27617.68202733401, 130.8150959041496]]}}}


@pytest.mark.parametrize("atmosphere", [WILD_SPLINE, OVERFLOWING_SPLINE], ids=["seed4899", "seed31148"])
@pytest.mark.parametrize("earth", ["FlatDistorted", "SimpleSphere"])
def test_pathological_atmosphere_is_still_bit_exact(gpu_ctx, oracle_det, earth, atmosphere):
    """Seed 4899 of the random sweep: a Natural spline through two knots 16 m apart swings to 36 K at 15.5 km and below 0 K beyond, the
    hydrostatic pressure reaches 1.7e308 Pa and then inf / NaN.  The GPU's shortcut divisions (dm_div ...: the IEEE quotient for operands
    with |exponent| < 500) returned NaN for n(h) where the host's IEEE division returns 1.0, and a ray that wandered into that zone
    took another path.  Since then the shortcuts run only inside altitude intervals in which atm_certify proves the operands tame, and
    IEEE operations everywhere else: the sampler, the integrator and whole frames must match the oracle to the bit (NaN as NaN)."""
    import ctypes as C
    from atm_raytracer_amd import config, generators
    from util import bits
    doc = {"view": {"position": {"latitude": 46.68895508040043, "longitude": 8.336055024224262, "altitude": {"Absolute": 3342.573230139556}},
                    "frame": {"direction": 180.0, "tilt": -20.804093573300673, "fov": 17.672070693357174, "max_distance": 50394.203235634996}},
           "earth_shape": earth, "straight_rays": False, "simulation_step": 50.0, "wavelength": 6.52274204561565e-07,
           "atmosphere": atmosphere, "scene": {"terrain_alpha": 1.0}, "output": {"width": 28, "height": 30, "generator": "Fast"}}
    cfg = config.Config.from_dict(doc)
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_params(gpu_ctx.handle, C.byref(cfg.params)))
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_atmosphere(gpu_ctx.handle, C.byref(cfg.atmosphere)))
    try:
        # the sampler across the tame part, the edge of the certificate (3.1 km), the overflow zone and the NaN zone; in runs of 64 equal
        # altitudes too, so that whole wavefronts sit inside one interval and take the certified path where there is one
        alt = np.concatenate([np.linspace(-600.0, 30000.0, 6121), np.linspace(3100.0, 3170.0, 1401), np.linspace(15499.2, 15499.4, 401),
                              np.repeat(np.linspace(-400.0, 17000.0, 120), 64)])
        got = generators.atmosphere_sample(gpu_ctx, alt)
        env = oracle_det.env(cfg.atmosphere, cfg.params.wavelength)
        want = {"temperature": [oracle_det.temperature(env, h) for h in alt], "pressure": [oracle_det.pressure(env, h) for h in alt],
                "n": [oracle_det.n(env, h) for h in alt], "dn_dh": [oracle_det.dn(env, h) for h in alt]}
        for k in want:
            bad = np.flatnonzero(bits(got[k]) != bits(np.array(want[k], dtype=np.float64)))
            assert bad.size == 0, (k, bad.size, alt[bad[:4]], got[k][bad[:4]], np.array(want[k])[bad[:4]])
        assert np.isnan(want["n"]).any() and np.isfinite(want["n"]).any()
        assert (np.array(want["pressure"]) > 1e300).any() or not np.isfinite(want["pressure"]).all()
        # the integrator: rays that dive into the wild zone (the three of the failing row and its neighbours) and rays that do not
        ang = np.array([-26.48440201045119, -27.1, -25.8, -12.0, -2.0, 0.0, 1.5, 20.0, 60.0])
        x, h = generators.ray_paths(gpu_ctx, 3342.573230139556, ang, 50.0, 1100, False)
        xo, ho = oracle_det.ray_paths(cfg.params, 3342.573230139556, ang, 50.0, 1100, False, cfg.atmosphere)
        assert np.array_equal(bits(x), bits(xo)) and np.array_equal(bits(h), bits(ho))
        assert np.isnan(ho).any()
        # whole frames, every generator
        tiles = synth.synth_tiles([46], [8], level=301)
        for gen, w, hh in (("Fast", 28, 30), ("Rectilinear", 28, 24), ("InterpolatingRectilinear", 28, 30)):
            cfg.params.generator = {"Fast": 0, "InterpolatingRectilinear": 1, "Rectilinear": 2}[gen]  # atmrt_generator_kind
            cfg.params.width, cfg.params.height = w, hh
            assert_bitexact(run_gpu(gpu_ctx, cfg, tiles), run_oracle(oracle_det, cfg, tiles))
    finally:
        us = config.us76()
        gpu_ctx.check(gpu_ctx.lib.atmrt_set_atmosphere(gpu_ctx.handle, C.byref(us)))


UNCERTIFIED = [  # earth model, step, max_distance, observer height above the terrain, tilt
    ({"Spherical": {"radius": 500.0}}, 0.02, 4.0, 2.0, -20.0),
    ({"Spherical": {"radius": 3.0e13}}, 50.0, 6_000.0, 30.0, -6.0),
    ("SimpleSphere", 2.0e-4, 0.05, 0.01, -30.0),
    ({"ObserverAe": {"proj_radius": 1.0e-40}}, 50.0, 6_000.0, 30.0, -6.0),
]


def uncertified_case(case, generator):
    from atm_raytracer_amd import config
    earth, step, max_distance, above, tilt = case
    doc = {"view": {"position": {"latitude": 46.5, "longitude": 8.5, "altitude": {"Relative": above}},
                    "frame": {"direction": 100.0, "tilt": tilt, "fov": 40.0, "max_distance": max_distance}},
           "earth_shape": earth, "simulation_step": step, "output": {"width": 24, "height": 16, "generator": generator}}
    return config.Config.from_dict(doc)


@pytest.mark.parametrize("case", UNCERTIFIED, ids=["radius_500m", "radius_3e13m", "step_0.2mm", "proj_radius_1e-40m"])
@pytest.mark.parametrize("generator", ["Fast", "Rectilinear"])
def test_uncertified_geometry_runs_on_ieee_operations(gpu_ctx, oracle_det, case, generator):
    """atm_certify certifies nothing for a planet below 1 km or above 1e12 m or a step below 1 mm, and x / calc_radius leaves dm_div for
    radii outside 1e-30 .. 1e30: these frames run the generic evaluation (IEEE division and square root in every stage) from the first
    step to the last and must still match the oracle bit for bit."""
    cfg = uncertified_case(case, generator)
    tiles = synth.synth_tiles([46], [8], level=301)
    want = run_oracle(oracle_det, cfg, tiles)
    assert want["n_hits"] > 0
    assert_bitexact(run_gpu(gpu_ctx, cfg, tiles), want)


def _nested_cylinders(cfg, n, lat=46.53, lon=8.5, r0=4.0, dr=3.0, alpha=0.5):
    """n concentric translucent cylinders on one spot: a ray through them collects up to 2 n points inside ONE 100 m step."""
    from atm_raytracer_amd import _abi
    objs = []
    for i in range(n):
        o = _abi.Object()
        o.kind, o.r1, o.r2, o.height = _abi.OBJ_FRUSTUM, r0 + dr * i, r0 + dr * i, 900.0
        o.position.latitude, o.position.longitude = lat, lon
        o.position.altitude_kind, o.position.altitude = _abi.ALT_RELATIVE, 0.0
        o.color[0], o.color[1], o.color[2], o.color[3] = 0.1 + 0.04 * i, 0.9 - 0.04 * i, 0.5, alpha
        objs.append(o)
    cfg.objects = objs
    return cfg


@pytest.mark.parametrize("generator,w,h", [("Rectilinear", 40, 24), ("Fast", 64, 32), ("InterpolatingRectilinear", 64, 32)])
def test_more_trace_points_in_one_step_than_the_step_list(gpu_ctx, oracle_det, generator, w, h):
    """20 translucent cylinders nested inside one another 3.3 km ahead: the steps that cross them produce up to 41 trace points
    (step_result of the reference has no bound, utils.rs:213-282) against the 12 the device keeps in registers — those steps are
    sorted in HBM (big_step_sort); an interpolating pixel's four lattice corners then hold well over a hundred points (k_interp_blend_big).
    Round 1 answered both with ATMRT_ERR_UNSUPPORTED."""
    from util import frame_stats
    cfg, tiles = synth.scene("S2", w, h, generator=generator, terrain_alpha=0.5, max_distance=20_000.0, tilt=3.0, fov=8.0)
    _nested_cylinders(cfg, 20)
    got = run_gpu(gpu_ctx, cfg, tiles)
    stats = frame_stats(gpu_ctx)
    assert stats["big_steps"] > 0, stats
    if generator == "InterpolatingRectilinear":  # the blend merges same-class points within one step's distance: few points per pixel
        assert stats["big_blend_pixels"] > 0, stats
    else:
        assert got["hit_count"].max() > 24, int(got["hit_count"].max())
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))
    # and with an opaque core: the step ends the march, every point of the step is still reported in prop order (utils.rs:279-285)
    cfg.objects[0].color[3] = 1.0
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert frame_stats(gpu_ctx)["big_steps"] > 0
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))


@pytest.mark.parametrize("alpha", [1.0, 0.5], ids=["opaque", "translucent"])
@pytest.mark.parametrize("case", ["all-down", "all-up", "one-step-past-a-slice", "exactly-one-slice", "ragged-groups"])
def test_time_sliced_march_edges(gpu_ctx, oracle_det, case, alpha):
    """Corner cases of the time-sliced Rectilinear march (csrc/atmrt_march_impl.h, k_rect_march_first / _cont; every test frame of
    more than 128 steps per ray takes it): every ray ends inside the first slice (no FIFO entry is ever written, the second kernel is
    not launched), no ray ends before the last slice, rays of 129 steps (one step in the second slice) and of 128 (not sliced at
    all: the plain kernel, for comparison of the boundary), and a frame whose pixel count is neither a multiple of a wavefront nor
    of a workgroup."""
    kw = dict(generator="Rectilinear", terrain_alpha=alpha)
    if case == "all-down":
        cfg, tiles = synth.scene("S2", 48, 20, tilt=-70.0, fov=30.0, max_distance=60_000.0, **kw)
    elif case == "all-up":
        cfg, tiles = synth.scene("S2", 48, 20, tilt=45.0, fov=30.0, max_distance=40_000.0, **kw)
    elif case == "one-step-past-a-slice":
        cfg, tiles = synth.scene("S2", 40, 24, tilt=-1.0, max_distance=12_850.0, **kw)  # 129 samples of 100 m
    elif case == "exactly-one-slice":
        cfg, tiles = synth.scene("S2", 40, 24, tilt=-1.0, max_distance=12_550.0, **kw)  # 126 samples: n_t + 2 = 128, unsliced
    else:
        cfg, tiles = synth.scene("S2", 37, 11, tilt=-2.0, max_distance=50_000.0, **kw)
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))
    if case == "all-down":
        assert (got["hit_count"] > 0).all()
    if case == "all-up":
        assert got["n_hits"] == 0

"""The C ABI itself on the GPU: DTED directory loading (the library's own parser), error statuses in place of the
reference's panics, edge inputs (empty terrain, tiles with n_lat != n_lon, southern/western cells, ragged shards,
1-pixel images, rays that drop below -1000 m), harness entry points."""
import ctypes as C
import os

import numpy as np
import pytest

from atm_raytracer_amd import _abi, config, generators, synth
from atm_raytracer_amd._lib import AtmrtError
from util import assert_bitexact, run_gpu, run_oracle

pytestmark = pytest.mark.gpu


def test_terrain_from_folder_matches_add_tile(gpu_ctx, oracle_det, tmp_path):
    """Terrain::from_folder (terrain/mod.rs:66-83) through the library's DTED reader == tiles handed over directly."""
    cfg, tiles = synth.scene("S3", 96, 48, step=250.0, level=301)
    synth.write_terrain_dir(str(tmp_path / "terrain"), tiles)
    gpu_ctx.check(gpu_ctx.lib.atmrt_terrain_clear(gpu_ctx.handle))
    terrain = generators.Terrain.from_folder(str(tmp_path / "terrain"), gpu_ctx)
    assert terrain.n_files == 9
    got = generators.make_generator(generators.Params(cfg), terrain).generate()
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))
    # the oracle's own reader agrees as well
    t, n = oracle_det.terrain_load_dir(str(tmp_path / "terrain"))
    assert n == 9
    assert_bitexact(got, oracle_det.generate(cfg.params, cfg.atmosphere, t))
    oracle_det.terrain_free(t)


def test_terrain_folder_errors(gpu_ctx, tmp_path):
    """Missing directory -> ATMRT_ERR_IO ("Error opening the terrain data directory", terrain/mod.rs:70-71);
    a file that is not a terrain tile -> ATMRT_ERR_FORMAT ("Could not buffer terrain file", :113-118)."""
    with pytest.raises(AtmrtError) as e:
        generators.Terrain.from_folder(str(tmp_path / "nope"), gpu_ctx)
    assert e.value.status == _abi.ERR_IO and "terrain data directory" in e.value.message
    d = tmp_path / "bad"
    d.mkdir()
    (d / "notes.txt").write_text("hello")
    with pytest.raises(AtmrtError) as e:
        generators.Terrain.from_folder(str(d), gpu_ctx)
    assert e.value.status == _abi.ERR_FORMAT and "Could not buffer terrain file" in e.value.message
    synth.write_dted(str(d / "trunc.dt1"), 46, 8, synth.synth_tile(46, 8, level=31))
    os.truncate(str(d / "trunc.dt1"), 4000)
    os.remove(str(d / "notes.txt"))
    with pytest.raises(AtmrtError) as e:
        generators.Terrain.from_folder(str(d), gpu_ctx)
    assert e.value.status == _abi.ERR_FORMAT


def test_parameter_validation(gpu_ctx):
    cfg, _ = synth.scene("S1", 16, 8)
    bad = [("simulation_step", 0.0), ("simulation_step", float("nan")), ("wavelength", -1.0), ("width", 0), ("generator", 7)]
    for field, value in bad:
        p = _abi.Params.from_buffer_copy(cfg.params)
        setattr(p, field, value)
        assert gpu_ctx.lib.atmrt_set_params(gpu_ctx.handle, C.byref(p)) == _abi.ERR_INVALID_ARGUMENT, field
        assert gpu_ctx.lib.atmrt_last_error(gpu_ctx.handle)
    p = _abi.Params.from_buffer_copy(cfg.params)
    p.frame.max_distance = float("inf")
    assert gpu_ctx.lib.atmrt_set_params(gpu_ctx.handle, C.byref(p)) == _abi.ERR_INVALID_ARGUMENT
    p = _abi.Params.from_buffer_copy(cfg.params)
    p.col_begin, p.col_end = 10, 4
    assert gpu_ctx.lib.atmrt_set_params(gpu_ctx.handle, C.byref(p)) == _abi.ERR_INVALID_ARGUMENT
    p = _abi.Params.from_buffer_copy(cfg.params)
    p.earth.kind = 99
    assert gpu_ctx.lib.atmrt_set_params(gpu_ctx.handle, C.byref(p)) == _abi.ERR_INVALID_ARGUMENT
    a = config.us76()
    a.n_functions = 0
    assert gpu_ctx.lib.atmrt_set_atmosphere(gpu_ctx.handle, C.byref(a)) == _abi.ERR_INVALID_ARGUMENT
    o = _abi.Object()
    o.kind = _abi.OBJ_BILLBOARD  # no texture
    assert gpu_ctx.lib.atmrt_objects_set(gpu_ctx.handle, C.byref(o), 1) == _abi.ERR_INVALID_ARGUMENT
    assert gpu_ctx.lib.atmrt_objects_set(gpu_ctx.handle, None, 0) == 0
    assert gpu_ctx.lib.atmrt_generate(gpu_ctx.handle, None) == _abi.ERR_INVALID_ARGUMENT


@pytest.mark.parametrize("generator", ["Fast", "Rectilinear", "InterpolatingRectilinear"])
def test_empty_terrain_and_relative_observer(gpu_ctx, oracle_det, generator):
    """No tile at all: every lookup is None -> 0.0 (utils.rs:28-31,84); a Relative observer then stands 1 m above sea level
    (Config::default position, params.rs:42-54)."""
    cfg = config.Config.from_dict({"output": {"width": 40, "height": 24, "generator": generator}, "simulation_step": 100.0,
                                   "view": {"frame": {"max_distance": 20000.0, "tilt": -1.0}}})
    assert_bitexact(run_gpu(gpu_ctx, cfg, {}), run_oracle(oracle_det, cfg, {}))


def test_nonsquare_southern_western_tiles(gpu_ctx, oracle_det):
    """Cells south of the equator / west of Greenwich (floor of negative coordinates, terrain/mod.rs:121-122) with fewer
    longitude than latitude posts (DTED zones above 50 degrees latitude)."""
    rng = np.random.default_rng(21)
    tiles = {}
    for la in (-34, -33):
        for lo in (-71, -70):
            base = synth.synth_tile(46, 8, level=241).astype(np.int32)[:, ::2][:, :121] + int(rng.integers(0, 300))
            tiles[(la, lo)] = base.astype(np.int16)
    assert tiles[(-34, -71)].shape == (241, 121)
    cfg = config.Config.from_dict({"view": {"position": {"latitude": -33.02, "longitude": -70.01, "altitude": {"Relative": 800.0}},
                                            "frame": {"direction": 200.0, "fov": 70.0, "tilt": -6.0, "max_distance": 90000.0}},
                                   "simulation_step": 150.0, "output": {"width": 80, "height": 40}})
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert got["n_hits"] > 500
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))
    cfg.params.generator = _abi.GENERATORS["Rectilinear"]
    cfg.params.width, cfg.params.height = 40, 20
    assert_bitexact(run_gpu(gpu_ctx, cfg, tiles), run_oracle(oracle_det, cfg, tiles))


@pytest.mark.parametrize("w,h,shard", [(1, 1, None), (1, 37, None), (65, 1, None), (130, 67, (63, 129)), (257, 33, (0, 1))])
def test_odd_sizes_and_ragged_shards(gpu_ctx, oracle_det, w, h, shard):
    """Image sizes that are not multiples of the 64-column x 32-row tile of k_fast_intersect, single rows/columns, and
    shards that start and end inside a wavefront."""
    for gen in ("Fast", "Rectilinear"):
        cfg, tiles = synth.scene("S2", w, h, generator=gen, max_distance=40_000.0, tilt=-3.0, level=301)
        if shard:
            cfg.params.col_begin, cfg.params.col_end = shard
        assert_bitexact(run_gpu(gpu_ctx, cfg, tiles), run_oracle(oracle_det, cfg, tiles))


def test_rays_below_minus_1000_and_short_paths(gpu_ctx, oracle_det):
    """A camera looking steeply down from altitude over empty terrain on a flat Earth with translucent terrain: rays cross
    sea level, keep going and end at -1000 m (utils.rs:167, rectilinear.rs:178), so paths are shorter than the terrain cache."""
    for gen in ("Fast", "Rectilinear"):
        cfg, _ = synth.scene("S1", 48, 32, generator=gen, earth_shape="FlatDistorted", tilt=-35.0, terrain_alpha=0.25,
                             max_distance=30_000.0)
        got = run_gpu(gpu_ctx, cfg, {})
        want = run_oracle(oracle_det, cfg, {})
        assert want["ray_steps"] < 48 * 32 * 299  # the paths really are truncated
        assert_bitexact(got, want)


def test_device_planes_equal_host_result(gpu_ctx):
    """atmrt_generate_device (caller-owned HBM planes, what bench.py times) == atmrt_generate (host SoA)."""
    import torch
    cfg, tiles = synth.scene("S2", 96, 40, terrain_alpha=0.5, tilt=-4.0)
    for gen in ("Fast", "Rectilinear"):
        cfg.params.generator = _abi.GENERATORS[gen]
        host = run_gpu(gpu_ctx, cfg, tiles)
        terrain = generators.Terrain(gpu_ctx)
        g = generators.make_generator(generators.Params(cfg), terrain)
        dev = torch.device("cuda", 0)
        h, w = 40, 96
        t = {k: torch.full((h, w), -1.0, dtype=torch.float64, device=dev) for k in ("azimuth", "elevation_angle", "lat", "lon", "distance", "elevation", "path_length")}
        t["normal"] = torch.zeros((3, h, w), dtype=torch.float64, device=dev)
        t["hit_count"] = torch.zeros((h, w), dtype=torch.int32, device=dev)
        steps, ms = g.generate_device(_abi.DevicePlanes(**{k: v.data_ptr() for k, v in t.items()}))
        torch.cuda.synchronize()
        assert steps == host["ray_steps"] and ms > 0
        assert np.array_equal(t["hit_count"].cpu().numpy().astype(np.uint32), host["hit_count"])
        assert np.array_equal(t["azimuth"].cpu().numpy(), host["azimuth"])
        has = host["hit_count"] > 0
        first = host["hit_offset"][has].astype(np.int64)
        for k in ("lat", "lon", "distance", "elevation", "path_length"):
            plane = t[k].cpu().numpy()
            assert np.array_equal(plane[has], host[k][first]) and np.isnan(plane[~has]).all()
        nrm = t["normal"].cpu().numpy()
        assert np.array_equal(np.stack([nrm[i][has] for i in range(3)], axis=1), host["normal"][first])
        tm = g.last_timings()
        assert tm["ray_steps"] == steps and tm["total_ms"] > 0


def test_atmosphere_harness_and_custom_layers(gpu_ctx, oracle_det):
    """output-atm (atm_printer.rs:37-46) on the device, US-76 and a custom two-layer Linear atmosphere from YAML."""
    doc = {"atmosphere": {"pressure": {"altitude": 100.0, "pressure": 100000.0}, "first_temperature_function": {"Linear": {"gradient": -0.008}},
                          "next_functions": [{"altitude": 2000.0, "function": {"Linear": {"gradient": 0.002}}},
                                             {"altitude": 9000.0, "function": {"Linear": {"gradient": 0.0}}}],
                          "temperature_fixed_point": {"altitude": 2500.0, "temperature": 270.0}},
           "output": {"width": 48, "height": 32}, "view": {"frame": {"max_distance": 50000.0, "tilt": -0.5}}, "simulation_step": 100.0}
    cfg = config.Config.from_dict(doc)
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_params(gpu_ctx.handle, C.byref(cfg.params)))
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_atmosphere(gpu_ctx.handle, C.byref(cfg.atmosphere)))
    alt = np.concatenate([np.linspace(-500, 30000, 500), [1999.99, 2000.0, 9000.0]])
    got = generators.atmosphere_sample(gpu_ctx, alt)
    env = oracle_det.env(cfg.atmosphere, cfg.params.wavelength)
    for i, h in enumerate(alt):
        assert got["temperature"][i] == oracle_det.temperature(env, h) and got["pressure"][i] == oracle_det.pressure(env, h)
        assert got["n"][i] == oracle_det.n(env, h) and got["dn_dh"][i] == oracle_det.dn(env, h)
    assert oracle_det.pressure(env, 100.0) == pytest.approx(100000.0, rel=1e-14)
    assert_bitexact(run_gpu(gpu_ctx, cfg, {}), run_oracle(oracle_det, cfg, {}))
    # restore the default atmosphere for the tests that follow
    us = config.us76()
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_atmosphere(gpu_ctx.handle, C.byref(us)))


def test_maximum_width(gpu_ctx, oracle_det):
    """The widest image the reference's i16 pixel arithmetic allows (fast.rs:116,122): 32767 columns."""
    cfg, tiles = synth.scene("S2", 32767, 2, max_distance=5_000.0, fov=90.0, tilt=-10.0, level=301)
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert got["hit_count"].shape == (2, 32767) and got["n_hits"] > 30000
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))
    p = _abi.Params.from_buffer_copy(cfg.params)
    p.width = 32768  # `x as i16` would wrap in the reference; rejected here
    assert gpu_ctx.lib.atmrt_set_params(gpu_ctx.handle, C.byref(p)) == _abi.ERR_INVALID_ARGUMENT


def test_two_contexts_compute_the_two_halves(oracle_det):
    """Contexts are independent (one per rank in the multi-GPU layout): two of them on one device, each holding the terrain and
    computing one pixel-column tile, together reproduce the whole frame."""
    cfg, tiles = synth.scene("S2", 96, 40, terrain_alpha=0.5, tilt=-4.0, level=301)
    want = run_oracle(oracle_det, cfg, tiles)
    ctxs = [generators.Context(0), generators.Context(0)]
    parts = []
    for ctx, (c0, c1) in zip(ctxs, ((0, 37), (37, 96))):
        terrain = generators.Terrain.from_tiles(tiles, ctx)
        p = generators.Params(cfg)
        p.pod = _abi.Params.from_buffer_copy(cfg.params)
        p.pod.col_begin, p.pod.col_end = c0, c1
        parts.append(generators.make_generator(p, terrain).generate())
    for ctx in ctxs:
        ctx.close()
    for k in ("azimuth", "elevation_angle", "hit_count"):
        assert np.array_equal(np.concatenate([r[k] for r in parts], axis=1), want[k]), k
    assert parts[0]["ray_steps"] + parts[1]["ray_steps"] == want["ray_steps"]
    assert parts[0]["n_hits"] + parts[1]["n_hits"] == want["n_hits"]


def test_last_hits_device_matches_host_result(gpu_ctx):
    """atmrt_last_hits_device: the trace-point lists left in HBM by atmrt_generate_device (what a multi-GPU host gathers for
    translucent / object scenes) are those atmrt_generate returns; opaque frames have none and say so."""
    import torch
    cfg, tiles = synth.scene("S2", 64, 32, generator="Rectilinear", terrain_alpha=0.5, tilt=-4.0)
    want = run_gpu(gpu_ctx, cfg, tiles)
    g = generators.make_generator(generators.Params(cfg), generators.Terrain(gpu_ctx))
    dev = torch.device("cuda", 0)
    h, w = 32, 64
    t = {k: torch.zeros((h, w), dtype=torch.float64, device=dev) for k in ("azimuth", "elevation_angle", "lat", "lon", "distance", "elevation", "path_length")}
    t["normal"] = torch.zeros((3, h, w), dtype=torch.float64, device=dev)
    t["hit_count"] = torch.zeros((h, w), dtype=torch.int32, device=dev)
    g.generate_device(_abi.DevicePlanes(**{k: v.data_ptr() for k, v in t.items()}))
    hits = g.last_hits_device(h, w)
    assert hits["lat"].shape[0] == want["n_hits"] > 0
    assert np.array_equal(hits["hit_offset"].cpu().numpy(), want["hit_offset"].astype(np.int64))
    for k in ("lat", "lon", "distance", "elevation", "path_length", "normal", "rgba"):
        assert np.array_equal(hits[k].cpu().numpy(), want[k]), k
    assert np.array_equal(hits["color_tag"].cpu().numpy().astype(np.uint32), want["color_tag"])
    # too small a capacity is refused; an opaque frame has no list
    n = C.c_uint64()
    pod = _abi.DeviceHits(capacity=want["n_hits"] - 1, **{k: v.data_ptr() for k, v in hits.items()})
    assert gpu_ctx.lib.atmrt_last_hits_device(gpu_ctx.handle, C.byref(pod), C.byref(n)) == _abi.ERR_INVALID_ARGUMENT and n.value == want["n_hits"]
    cfg.params.terrain_alpha = 1.0
    g2 = generators.make_generator(generators.Params(cfg), generators.Terrain(gpu_ctx))
    g2.generate_device(_abi.DevicePlanes(**{k: v.data_ptr() for k, v in t.items()}))
    assert gpu_ctx.lib.atmrt_last_hits_device(gpu_ctx.handle, None, C.byref(n)) == _abi.ERR_STATE


def test_failed_frame_invalidates_the_last_frame(gpu_ctx):
    """After a frame that fails, atmrt_draw_image / atmrt_last_hits_device must refuse with ATMRT_ERR_STATE instead of reading
    the (reused or reallocated) buffers of the frame before it."""
    from atm_raytracer_amd import generators
    cfg, tiles = synth.scene("S2", 64, 32, generator="InterpolatingRectilinear", terrain_alpha=0.5)
    run_gpu(gpu_ctx, cfg, tiles)  # a good frame with packed trace points
    col = generators.into_coloring(gpu_ctx.lib, cfg.params, cfg.coloring)
    rgb = np.zeros((32, 64, 3), dtype=np.uint8)
    gpu_ctx.check(gpu_ctx.lib.atmrt_draw_image(gpu_ctx.handle, C.byref(col), rgb.ctypes.data))
    gpu_ctx.check(gpu_ctx.lib.atmrt_debug_fail_next_frame(gpu_ctx.handle))
    with pytest.raises(AtmrtError) as e:
        run_gpu(gpu_ctx, cfg, tiles)
    assert e.value.status == _abi.ERR_HIP and "injected" in e.value.message
    assert gpu_ctx.lib.atmrt_draw_image(gpu_ctx.handle, C.byref(col), rgb.ctypes.data) == _abi.ERR_STATE
    n = C.c_uint64()
    assert gpu_ctx.lib.atmrt_last_hits_device(gpu_ctx.handle, None, C.byref(n)) == _abi.ERR_STATE
    run_gpu(gpu_ctx, cfg, tiles)  # and the context recovers
    gpu_ctx.check(gpu_ctx.lib.atmrt_draw_image(gpu_ctx.handle, C.byref(col), rgb.ctypes.data))

"""The multi-GPU path below the C ABI (csrc/atmrt_multi.hip), rehearsed on the ONE GPU of the test box: a device may be
listed more than once in atmrt_ctx_create_multi, so [0, 0, 0] is three sub-contexts with three host threads that cut every frame
into three pixel-column tiles — exactly the code an 8-GPU node runs, with device-to-device copies in place of xGMI.  RCCL refuses
two ranks on one device, so its route runs at world size 1 (same calls: ncclCommInitAll / ncclCommInitRank, ncclAllGather).

The bar: what a multi-device context returns IS the single-context frame — every plane, every trace point, every offset,
bit for bit, for all three generators, with and without variable-length lists, also when the width does not divide."""
import ctypes as C
import threading

import numpy as np
import pytest
import torch

from atm_raytracer_amd import _abi, generators, synth
from util import assert_bitexact, bits, run_gpu

pytestmark = pytest.mark.gpu

GENERATORS = ["Fast", "Rectilinear", "InterpolatingRectilinear"]


def scene(generator, lists, width=96, height=40):
    """A small S2 frame; `lists`: translucent terrain + a few objects, so pixels hold several trace points."""
    kw = dict(terrain_alpha=0.5, tilt=-4.0) if lists else {}
    cfg, tiles = synth.scene("S2", width, height, generator=generator, max_distance=60_000.0, **kw)
    if lists:
        synth.add_objects(cfg, n_cyl=14, n_bill=6, dist=(1_000.0, 40_000.0), spread_deg=25.0)
    return cfg, tiles


def dense_first_hits(res):
    """What atmrt_generate_device leaves in the planes: the first trace point of every pixel, NaN where there is none."""
    hc = res["hit_count"]
    first = res["hit_offset"].astype(np.int64)
    has = hc > 0
    out = {"azimuth": res["azimuth"], "elevation_angle": res["elevation_angle"], "hit_count": hc.astype(np.int32)}
    for k in ("lat", "lon", "distance", "elevation", "path_length"):
        plane = np.full(hc.shape, np.nan)
        plane[has] = res[k][first[has]]
        out[k] = plane
    nrm = np.full((3,) + hc.shape, np.nan)
    for c in range(3):
        nrm[c][has] = res["normal"][first[has], c]
    out["normal"] = nrm
    return out


def check_image(planes, want):
    ref = dense_first_hits(want)
    for k, v in ref.items():
        got = planes[k].cpu().numpy()
        assert got.shape == v.shape, k
        if k == "hit_count":
            assert np.array_equal(got, v), k
            continue
        g, w = bits(got), bits(v)
        if k not in ("azimuth", "elevation_angle"):  # planes are only defined where the pixel has a trace point
            sel = np.broadcast_to(ref["hit_count"] > 0, v.shape)
            g, w = g[sel], w[sel]
        assert np.array_equal(g, w), (k, int((g != w).sum()))


def check_lists(hits, want):
    assert np.array_equal(hits["hit_offset"].cpu().numpy().astype(np.uint64), want["hit_offset"])
    for k in ("lat", "lon", "distance", "elevation", "path_length", "normal", "rgba"):
        assert np.array_equal(bits(hits[k].cpu().numpy()), bits(want[k])), k
    assert np.array_equal(hits["color_tag"].cpu().numpy().astype(np.uint32), want["color_tag"])


@pytest.fixture(scope="module")
def multi3():
    ctx = generators.Context.multi([0, 0, 0])
    yield ctx
    ctx.close()


@pytest.mark.parametrize("lists", [False, True], ids=["opaque", "lists"])
@pytest.mark.parametrize("generator", GENERATORS)
def test_multi_context_host_frame_is_the_single_context_frame(gpu_ctx, multi3, generator, lists):
    """atmrt_generate on three sub-contexts: tiles of 33 / 34 / 34 columns (101 does not divide), planes copied straight into the
    one [H][W] block, lists merged row segment by row segment."""
    cfg, tiles = scene(generator, lists, width=101, height=37)
    want = run_gpu(gpu_ctx, cfg, tiles)
    got = run_gpu(multi3, cfg, tiles)
    assert got["hit_count"].shape == (37, 101) and want["n_hits"] > 0
    if lists:
        assert want["hit_count"].max() > 1 and (want["color_tag"] == 1).any()
    if generator == "InterpolatingRectilinear":
        # every tile computes its own angular lattice, so lattice pixels along a tile boundary are marched by both neighbours: the
        # image is identical, the work is slightly more (ray_steps counts the lattice pixels a tile references)
        assert want["ray_steps"] <= got["ray_steps"] <= 1.2 * want["ray_steps"]
        got["ray_steps"] = want["ray_steps"]
    assert_bitexact(got, want)
    tm = multi3.comm_timings()
    assert tm["world"] == 3 and tm["route"] == "host"
    # renderer compositing of the tiles into one host image
    conf = dict(kind=_abi.COLORING_SHADING, water_level=0.0, ambient_light=0.4, light_zenith_angle=45.0, light_dir=0.0,
                palette=_abi.PALETTES["Improved"], has_fog=1, fog_distance=40_000.0)
    col = generators.into_coloring(gpu_ctx.lib, cfg.params, conf)
    run_gpu(gpu_ctx, cfg, tiles)
    assert np.array_equal(generators.draw_image(multi3, col, 101, 37), generators.draw_image(gpu_ctx, col, 101, 37))


def make_ctx(kind):
    if kind == "peer-2":
        return generators.Context.multi([0, 0])
    if kind == "peer-5":
        return generators.Context.multi([0] * 5)
    if kind == "rccl-multi-1":
        return generators.Context.multi([0])
    if kind == "rccl-rank-1":
        ctx = generators.Context(0)
        ctx.comm_init_rank(ctx.comm_unique_id(), 0, 1)
        return ctx
    raise KeyError(kind)


@pytest.mark.parametrize("kind", ["peer-2", "peer-5", "rccl-multi-1", "rccl-rank-1"])
def test_image_in_hbm_on_every_device(gpu_ctx, kind):
    """atmrt_generate_image_device + atmrt_image_hits_device + atmrt_draw_image_gathered_device: slab all-gather (RCCL, or peer
    copies between sub-contexts of one device), permutation into [H][W] planes, count -> scan -> offset for the lists."""
    ctx = make_ctx(kind)
    n_dev = len(ctx.devices)
    try:
        for generator in GENERATORS:
            for lists in (False, True):
                cfg, tiles = scene(generator, lists, width=90 if n_dev != 5 else 93, height=33)
                W, H = cfg.params.width, cfg.params.height
                want = run_gpu(gpu_ctx, cfg, tiles)
                ctx.check(ctx.lib.atmrt_terrain_clear(ctx.handle))
                gen = generators.make_generator(generators.Params(cfg), generators.Terrain.from_tiles(tiles, ctx))
                images = [generators.image_planes(H, W, torch.device("cuda", 0)) for _ in range(n_dev)]
                steps, _ = gen.generate_image_device([pod for _, pod in images])
                assert steps == want["ray_steps"] or (generator == "InterpolatingRectilinear" and want["ray_steps"] <= steps <= 1.3 * want["ray_steps"])
                for planes, _ in images:
                    check_image(planes, want)
                tm = ctx.comm_timings()
                assert tm["world"] == n_dev and tm["route"] == ("rccl" if kind.startswith("rccl") else "peer") and tm["collectives"] == 1
                if lists or generator == "InterpolatingRectilinear":
                    hits = gen.image_hits_device(H, W)
                    for h in (hits if isinstance(hits, list) else [hits]):
                        check_lists(h, want)
                    assert ctx.comm_timings()["collectives"] == 2  # the image's slabs (which carry the totals) + the lists' blocks
                else:
                    n = C.c_uint64()
                    assert ctx.lib.atmrt_image_hits_device(ctx.handle, None, C.byref(n)) == _abi.ERR_STATE
                # the 3 B/pixel route: draw every tile, gather the RGB8 tiles
                conf = dict(kind=_abi.COLORING_SIMPLE, water_level=0.0, ambient_light=0.4, light_zenith_angle=45.0, light_dir=0.0,
                            palette=_abi.PALETTES["Legacy"], has_fog=0, fog_distance=1.0)
                col = generators.into_coloring(gpu_ctx.lib, cfg.params, conf)
                rgbs = [torch.zeros((H, W, 3), dtype=torch.uint8, device="cuda:0") for _ in range(n_dev)]
                ptrs = (C.c_void_p * n_dev)(*[t.data_ptr() for t in rgbs])
                ctx.check(ctx.lib.atmrt_draw_image_gathered_device(ctx.handle, C.byref(col), ptrs))
                run_gpu(gpu_ctx, cfg, tiles)
                ref = generators.draw_image(gpu_ctx, col, W, H)
                for t in rgbs:
                    assert np.array_equal(t.cpu().numpy(), ref)
    finally:
        ctx.close()


def test_external_transport_two_ranks_in_two_threads(gpu_ctx):
    """atmrt_ctx_comm_init_external: two rank contexts on the one GPU, driven by two host threads; the host's all-gather is a
    test double (a barrier and a shared buffer).  The path an MPI or gloo host takes."""
    return _two_rank_threads(gpu_ctx, "Rectilinear", True)


def test_external_transport_fast_generator(gpu_ctx):
    """The Fast generator (opaque frame: first-hit planes only, no lists) through the host transport, two rank threads, an odd
    number of rows."""
    return _two_rank_threads(gpu_ctx, "Fast", False)


def _two_rank_threads(gpu_ctx, generator, lists):
    world = 2
    barrier = threading.Barrier(world)
    shared = {}
    lock = threading.Lock()

    def transport(rank):
        def all_gather(send, recv):
            n = len(send)
            with lock:
                shared[rank] = bytes(send)
            barrier.wait()
            for r in range(world):
                C.memmove(C.addressof(recv) + r * n, shared[r], n)
            barrier.wait()
            if rank == 0:
                shared.clear()
            barrier.wait()
        return all_gather

    cfg, tiles = scene(generator, lists, width=70, height=31)
    W, H = cfg.params.width, cfg.params.height
    want = run_gpu(gpu_ctx, cfg, tiles)
    results, errors = {}, []

    def rank_main(rank):
        try:
            ctx = generators.Context(0)
            ctx.comm_init_external(rank, world, transport(rank))
            cfg_r, _ = scene(generator, lists, width=70, height=31)
            gen = generators.make_generator(generators.Params(cfg_r), generators.Terrain.from_tiles(tiles, ctx))
            planes, pod = generators.image_planes(H, W, torch.device("cuda", 0))
            steps, _ = gen.generate_image_device(pod)
            hits = gen.image_hits_device(H, W) if lists else None
            results[rank] = (planes, hits, steps, ctx.comm_timings())
            ctx.close()
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)
            barrier.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert sum(results[r][2] for r in range(world)) == want["ray_steps"]
    for r in range(world):
        planes, hits, _, tm = results[r]
        check_image(planes, want)
        if lists:
            check_lists(hits, want)
        assert tm["route"] == "external" and tm["world"] == 2


@pytest.mark.parametrize("generator", GENERATORS)
def test_multi_context_degenerate_frames(gpu_ctx, multi3, generator):
    """Frames at the edges of the tiling: a sky-only view (no trace point anywhere: empty lists on every device) and an image exactly
    as wide as the number of devices (one pixel column per tile)."""
    sky, tiles = synth.scene("S2", 64, 20, generator=generator, tilt=60.0, fov=20.0, max_distance=30_000.0)
    want = run_gpu(gpu_ctx, sky, tiles)
    assert want["n_hits"] == 0
    got = run_gpu(multi3, sky, tiles)
    got["ray_steps"] = want["ray_steps"] if generator == "InterpolatingRectilinear" else got["ray_steps"]
    assert_bitexact(got, want)
    thin, tiles = synth.scene("S2", 3, 17, generator=generator, tilt=-3.0, max_distance=40_000.0, terrain_alpha=0.5)
    want = run_gpu(gpu_ctx, thin, tiles)
    assert want["n_hits"] > 0
    got = run_gpu(multi3, thin, tiles)
    got["ray_steps"] = want["ray_steps"] if generator == "InterpolatingRectilinear" else got["ray_steps"]
    assert_bitexact(got, want)
    images = [generators.image_planes(17, 3, torch.device("cuda", 0)) for _ in range(3)]
    gen = generators.make_generator(generators.Params(thin), generators.Terrain.from_tiles(tiles, multi3))
    gen.generate_image_device([pod for _, pod in images])
    for planes, _ in images:
        check_image(planes, want)
    for h in gen.image_hits_device(17, 3):
        check_lists(h, want)


def test_multi_context_errors_and_recovery(gpu_ctx, multi3):
    cfg, tiles = scene("Fast", False, width=60, height=24)
    # the library assigns the tiles: a caller's own column shard is refused
    cfg.params.col_begin, cfg.params.col_end = 0, 30
    assert multi3.lib.atmrt_set_params(multi3.handle, C.byref(cfg.params)) == _abi.ERR_INVALID_ARGUMENT
    cfg.params.col_begin = cfg.params.col_end = 0
    # fewer columns than devices
    narrow, _ = scene("Fast", False, width=2, height=8)
    with pytest.raises(generators.AtmrtError) as e:
        run_gpu(multi3, narrow, tiles)
    assert "less than" in str(e.value)
    # a frame that fails on one device fails as a whole, names the device, and the context goes on
    want = run_gpu(gpu_ctx, cfg, tiles)
    multi3.check(multi3.lib.atmrt_debug_fail_next_frame(multi3.handle))
    with pytest.raises(generators.AtmrtError) as e:
        run_gpu(multi3, cfg, tiles)
    assert "device" in str(e.value) and "injected" in str(e.value)
    assert_bitexact(run_gpu(multi3, cfg, tiles), want)
    # the same on the device route: the frame fails before any exchange starts (nobody waits for the device that failed) ...
    gen = generators.make_generator(generators.Params(cfg), generators.Terrain.from_tiles(tiles, multi3))
    images = [generators.image_planes(24, 60, torch.device("cuda", 0)) for _ in range(3)]
    multi3.check(multi3.lib.atmrt_debug_fail_next_frame(multi3.handle))
    with pytest.raises(generators.AtmrtError) as e:
        gen.generate_image_device([pod for _, pod in images])
    assert "injected" in str(e.value)
    gen.generate_image_device([pod for _, pod in images])  # ... and the next frame is whole again
    for planes, _ in images:
        check_image(planes, want)
    # the device-only entry points of a plain context are refused on a multi-device one
    planes, pod = generators.image_planes(24, 60, torch.device("cuda", 0))
    assert multi3.lib.atmrt_generate_device(multi3.handle, C.byref(pod), None, None) == _abi.ERR_STATE
    # the diagnostic entry points run on the first device
    alt = np.array([0.0, 1000.0, 12000.0])
    a, b = generators.atmosphere_sample(multi3, alt), generators.atmosphere_sample(gpu_ctx, alt)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    assert multi3.lib.atmrt_ctx_device_count(multi3.handle) == 3 and gpu_ctx.lib.atmrt_ctx_device_count(gpu_ctx.handle) == 1


def _lists_scene(width=90, height=33, generator="Rectilinear"):
    return scene(generator, True, width=width, height=height)


def test_partially_null_destinations_do_not_hang(gpu_ctx, multi3):
    """atmrt_image_hits_device with dst = [pods0, NULL entry, pods2].  Every device takes part in the lists' collective
    whatever the caller wants on it; the skipped device gets nothing; a device whose own arguments are wrong (capacity too small)
    reports that AFTER taking part, and the others still get their lists."""
    cfg, tiles = _lists_scene()
    W, H = cfg.params.width, cfg.params.height
    want = run_gpu(gpu_ctx, cfg, tiles)
    multi3.check(multi3.lib.atmrt_terrain_clear(multi3.handle))
    gen = generators.make_generator(generators.Params(cfg), generators.Terrain.from_tiles(tiles, multi3))
    images = [generators.image_planes(H, W, torch.device("cuda", 0)) for _ in range(3)]
    gen.generate_image_device([pod for _, pod in images])
    hits = gen.image_hits_device(H, W, skip=(1,))
    assert hits[1] is None
    check_lists(hits[0], want)
    check_lists(hits[2], want)
    assert multi3.comm_timings()["collectives"] == 2
    # the total needs no collective and can be asked any number of times, from a multi-device context ...
    n = C.c_uint64()
    for _ in range(3):
        multi3.check(multi3.lib.atmrt_image_hits_device(multi3.handle, None, C.byref(n)))
        assert n.value == want["n_hits"]
    # ... a device whose capacity is too small fails alone, after the collective: the call reports it, the other devices hold the lists
    ts = [generators._hit_tensors(want["n_hits"], H, W, torch.device("cuda", 0)) for _ in range(3)]
    pods = (_abi.DeviceHits * 3)(*[_abi.DeviceHits(capacity=(5 if i == 1 else want["n_hits"]), **{k: v.data_ptr() for k, v in t.items()})
                                   for i, t in enumerate(ts)])
    assert multi3.lib.atmrt_image_hits_device(multi3.handle, pods, None) == _abi.ERR_INVALID_ARGUMENT
    assert "capacity" in multi3.lib.atmrt_last_error(multi3.handle).decode()
    check_lists(ts[0], want)
    check_lists(ts[2], want)
    # an image that was not assembled on a device (azimuth NULL) cannot have its lists there — reported, not hung
    pods_img = [pod for _, pod in images]
    pods_img[2] = _abi.DevicePlanes()
    gen.generate_image_device(pods_img)
    hits = gen.image_hits_device(H, W, skip=(2,))
    check_lists(hits[0], want)
    pods = (_abi.DeviceHits * 3)(*[_abi.DeviceHits(capacity=want["n_hits"], **{k: v.data_ptr() for k, v in t.items()}) for t in ts])
    assert multi3.lib.atmrt_image_hits_device(multi3.handle, pods, None) == _abi.ERR_INVALID_ARGUMENT
    assert "not assembled" in multi3.lib.atmrt_last_error(multi3.handle).decode()
    check_lists(ts[1], want)


@pytest.mark.parametrize("kind", ["multi", "rank"])
def test_lists_of_a_frame_without_trace_points(gpu_ctx, multi3, kind):
    """translucent terrain in a sky-only view — the frame has lists, and they are empty.  torch.empty(0).data_ptr() is 0:
    NULL list arrays are fine when the image has no trace point, hit_offset comes back zero-filled, no collective runs."""
    sky, tiles = synth.scene("S2", 64, 20, generator="Rectilinear", tilt=60.0, fov=20.0, max_distance=30_000.0, terrain_alpha=0.5)
    ctx = multi3 if kind == "multi" else generators.Context(0)
    try:
        if kind == "rank":
            ctx.comm_init_rank(ctx.comm_unique_id(), 0, 1)
        ctx.check(ctx.lib.atmrt_terrain_clear(ctx.handle))
        gen = generators.make_generator(generators.Params(sky), generators.Terrain.from_tiles(tiles, ctx))
        images = [generators.image_planes(20, 64, torch.device("cuda", 0)) for _ in ctx.devices]
        gen.generate_image_device([pod for _, pod in images])
        hits = gen.image_hits_device(20, 64)
        for h in (hits if isinstance(hits, list) else [hits]):
            assert h["lat"].numel() == 0 and h["lat"].data_ptr() == 0
            assert int(h["hit_offset"].abs().sum()) == 0
        assert ctx.comm_timings()["collectives"] == 1
        for planes, _ in images:
            assert int(planes["hit_count"].sum()) == 0
    finally:
        if kind == "rank":
            ctx.close()


@pytest.mark.parametrize("nth", [1, 2], ids=["image-collective", "lists-collective"])
def test_a_failing_collective_fails_the_frame_on_every_device_and_the_next_is_whole(gpu_ctx, multi3, nth):
    """a COLLECTIVE (not a tile) fails on one device — injected where a refused ncclAllGather would return.  The
    peer route: the failing device releases the others at their barrier, the call returns the failure, nothing hangs, and the next
    frame (and its lists) is whole."""
    cfg, tiles = _lists_scene()
    W, H = cfg.params.width, cfg.params.height
    want = run_gpu(gpu_ctx, cfg, tiles)
    multi3.check(multi3.lib.atmrt_terrain_clear(multi3.handle))
    gen = generators.make_generator(generators.Params(cfg), generators.Terrain.from_tiles(tiles, multi3))
    images = [generators.image_planes(H, W, torch.device("cuda", 0)) for _ in range(3)]
    multi3.fail_next_collective(index=1, nth=nth)
    with pytest.raises(generators.AtmrtError) as e:
        gen.generate_image_device([pod for _, pod in images])
        gen.image_hits_device(H, W)
    assert "injected" in str(e.value) and "device" in str(e.value)
    if nth == 1:  # the frame is void: its lists cannot be asked for
        n = C.c_uint64()
        assert multi3.lib.atmrt_image_hits_device(multi3.handle, None, C.byref(n)) == _abi.ERR_STATE
    gen.generate_image_device([pod for _, pod in images])
    for planes, _ in images:
        check_image(planes, want)
    for h in gen.image_hits_device(H, W):
        check_lists(h, want)


def test_a_failing_collective_on_the_external_device_route(gpu_ctx):
    """The same with two rank contexts and the host's own device transport: rank 1's collective fails before its callback runs;
    rank 0's callback — a test double that waits for its peer — is released by the double's own abort and reports failure, so
    both ranks return an error; after the double is reset the next frame is whole on both."""
    world = 2
    barrier = threading.Barrier(world)
    shared, lock = {}, threading.Lock()
    cfg, tiles = _lists_scene(width=70, height=31)
    W, H = cfg.params.width, cfg.params.height
    want = run_gpu(gpu_ctx, cfg, tiles)

    dev = torch.device("cuda", 0)

    def transport(rank):
        def all_gather(send, recv, nbytes):
            mine = torch.as_tensor(generators._DeviceBytes(send, nbytes), device=dev)
            out = torch.as_tensor(generators._DeviceBytes(recv, nbytes * world), device=dev)
            with lock:
                shared[rank] = mine
            barrier.wait(timeout=60)
            for r in range(world):
                out[r * nbytes:(r + 1) * nbytes].copy_(shared[r])
            torch.cuda.synchronize(dev)
            barrier.wait(timeout=60)
        return all_gather

    results, errors, failures = {}, [], {}
    frame_gate = threading.Barrier(world)

    def rank_main(rank):
        try:
            ctx = generators.Context(0)
            ctx.comm_init_external_device(rank, world, transport(rank))
            cfg_r, _ = _lists_scene(width=70, height=31)
            gen = generators.make_generator(generators.Params(cfg_r), generators.Terrain.from_tiles(tiles, ctx))
            planes, pod = generators.image_planes(H, W, torch.device("cuda", 0))
            if rank == 1:
                ctx.fail_next_collective(nth=1)
            try:
                gen.generate_image_device(pod)
                failures[rank] = None
            except generators.AtmrtError as exc:
                failures[rank] = str(exc)
                if rank == 1:
                    barrier.abort()  # what a real transport's failure detector does for the peers of a dead rank
            frame_gate.wait(timeout=120)
            if rank == 0:
                barrier.reset()
            frame_gate.wait(timeout=120)
            gen.generate_image_device(pod)
            hits = gen.image_hits_device(H, W)
            results[rank] = (planes, hits)
            ctx.close()
        except Exception as exc:  # noqa: BLE001
            errors.append(exc)
            barrier.abort()
            frame_gate.abort()

    threads = [threading.Thread(target=rank_main, args=(r,)) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    assert "injected" in failures[1] and "callback returned" in failures[0]
    for r in range(world):
        check_image(results[r][0], want)
        check_lists(results[r][1], want)


@pytest.mark.parametrize("generator", GENERATORS)
def test_tile_widths_that_change_between_frames(gpu_ctx, generator):
    """the library owns the tiling and re-cuts it between frames.  Three frames of one context with three different
    tilings (equal; very unequal incl. one-column tiles; the library's own rule applied to made-up tile times): every frame is the
    single-context frame bit for bit, planes and lists, and atmrt_ctx_tile_columns reports what was used."""
    ctx = generators.Context.multi([0] * 4)
    try:
        cfg, tiles = scene(generator, True, width=97, height=29)
        W, H = cfg.params.width, cfg.params.height
        want = run_gpu(gpu_ctx, cfg, tiles)
        ctx.check(ctx.lib.atmrt_terrain_clear(ctx.handle))
        gen = generators.make_generator(generators.Params(cfg), generators.Terrain.from_tiles(tiles, ctx))
        images = [generators.image_planes(H, W, torch.device("cuda", 0)) for _ in range(4)]
        equal = [g * W // 4 for g in range(5)]
        out = (C.c_int32 * 5)()
        ms = (C.c_double * 4)(10.0, 30.0, 5.0, 20.0)
        assert ctx.lib.atmrt_tiles_rebalance(W, 4, (C.c_int32 * 5)(*equal), ms, out) == 0
        for cols in (None, [0, 1, 2, 60, W], list(out), [0, 50, 51, 96, W], None):
            ctx.set_tiling(cols)
            gen.generate_image_device([pod for _, pod in images])
            used = [ctx.tile_columns(i) for i in range(4)]
            assert [u[0] for u in used] + [used[-1][1]] == (cols or equal)
            for planes, _ in images:
                check_image(planes, want)
            for h in gen.image_hits_device(H, W):
                check_lists(h, want)
            # the host-consumer route cuts the same way
            got = gen.generate()
            if generator == "InterpolatingRectilinear":
                got["ray_steps"] = want["ray_steps"]
            assert_bitexact(got, want)
    finally:
        ctx.close()


def test_tiles_are_recut_from_the_tile_times(gpu_ctx, monkeypatch):
    """ATMRT_TILE_BALANCE=1 on sub-contexts of ONE device (where it is off by default: contended times mean nothing): whatever
    tilings the noisy times produce from frame to frame, every frame is the single-context frame."""
    monkeypatch.setenv("ATMRT_TILE_BALANCE", "1")
    ctx = generators.Context.multi([0] * 3)
    try:
        cfg, tiles = scene("Rectilinear", True, width=301, height=64)
        W, H = cfg.params.width, cfg.params.height
        want = run_gpu(gpu_ctx, cfg, tiles)
        ctx.check(ctx.lib.atmrt_terrain_clear(ctx.handle))
        gen = generators.make_generator(generators.Params(cfg), generators.Terrain.from_tiles(tiles, ctx))
        images = [generators.image_planes(H, W, torch.device("cuda", 0)) for _ in range(3)]
        seen = set()
        for frame in range(6):
            gen.generate_image_device([pod for _, pod in images])
            seen.add(tuple(ctx.tile_columns(i) for i in range(3)))
            for planes, _ in images:
                check_image(planes, want)
            for h in gen.image_hits_device(H, W):
                check_lists(h, want)
            got = gen.generate()  # the host route re-cuts from its own times too
            assert_bitexact(got, want)
        print(f"tilings used over 6 + 6 frames: {sorted(seen)}")
        assert all(t[0][0] == 0 and t[-1][1] == W and all(a[1] == b[0] and a[0] < a[1] for a, b in zip(t, t[1:])) for t in seen)
    finally:
        ctx.close()

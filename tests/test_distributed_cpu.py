"""The N > 1 path on the CPU: world_size 2 over gloo.  The product's exchange is C++ below the C ABI (csrc/atmrt_multi.hip:
tiles assigned by the library, one ncclAllGather of the slabs, k_assemble_image, count -> scan -> offset for the lists) and
cannot run without a GPU; what runs here is tests/sharding_model.py, the torch.distributed statement of the SAME layout — the
column tiles, the rank-major slab all-gather, the permutation into [H][W] planes, the list merge — with the oracle standing in
for the HIP library on each rank.  It is a model of the layout, kept as test infrastructure: nothing in the product imports
it.  The gathered image must equal the unsharded frame."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
PLANES = ("azimuth", "elevation_angle", "hit_count", "lat", "lon", "distance", "elevation", "path_length", "normal")


def dense_planes(res):
    """Dense first-hit planes like atmrt_generate_device writes them (NaN where the pixel has no trace point; normal planar)."""
    out = {"azimuth": res["azimuth"], "elevation_angle": res["elevation_angle"], "hit_count": res["hit_count"].astype(np.int32)}
    first = res["hit_offset"].astype(np.int64)
    has = res["hit_count"] > 0
    for name in ("lat", "lon", "distance", "elevation", "path_length"):
        plane = np.full(res["hit_count"].shape, np.nan)
        plane[has] = res[name][first[has]]
        out[name] = plane
    nrm = np.full((3,) + res["hit_count"].shape, np.nan)
    for c in range(3):
        nrm[c][has] = res["normal"][first[has], c]
    out["normal"] = nrm
    return out


def worker(rank, world, port, width, height, generator, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sharding_model as sharding
    from atm_raytracer_amd import synth
    from oracle_binding import Oracle
    from util import run_oracle
    cfg, tiles = synth.scene("S2", width, height, generator=generator, max_distance=60_000.0, level=301)
    c0, c1 = sharding.column_shard(width, rank, world)
    cfg.params.col_begin, cfg.params.col_end = c0, c1
    res = run_oracle(Oracle("det"), cfg, tiles, n_threads=2)
    # what bench.py does per frame: the shard's planes live in one slab (here filled from the oracle's shard), ONE
    # all_gather_into_tensor moves the slabs, and the same call leaves the [H][W] image — nothing happens after it
    slab = sharding.PlaneSlab(height, c1 - c0, torch.device("cpu"))
    for k, v in dense_planes(res).items():
        slab.planes[k].copy_(torch.from_numpy(np.ascontiguousarray(v)))
    gather = sharding.ImageGather(slab, world)
    calls = []
    real = dist.all_gather_into_tensor
    dist.all_gather_into_tensor = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    full = gather(dist)
    dist.all_gather_into_tensor = real
    assert len(calls) == 1, "one collective per frame"
    steps = torch.tensor([res["ray_steps"]], dtype=torch.int64)
    dist.all_reduce(steps)
    if rank == 0:
        queue.put(({k: v.numpy() for k, v in full.items()}, int(steps.item())))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("generator", ["Fast", "Rectilinear"])
def test_two_rank_column_shards_reassemble(generator, oracle_det):
    from atm_raytracer_amd import synth
    from util import run_oracle
    width, height, world = 32, 12, 2
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = 29500 + os.getpid() % 2000
    procs = [ctx.Process(target=worker, args=(r, world, port, width, height, generator, queue)) for r in range(world)]
    for p in procs:
        p.start()
    got, steps = queue.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg, tiles = synth.scene("S2", width, height, generator=generator, max_distance=60_000.0, level=301)
    want_res = run_oracle(oracle_det, cfg, tiles)
    want = dense_planes(want_res)
    assert steps == want_res["ray_steps"]
    for k in PLANES:
        assert got[k].shape == ((3, height, width) if k == "normal" else (height, width))
        assert np.array_equal(got[k], want[k], equal_nan=True), k


HIT_FIELDS = ("lat", "lon", "distance", "elevation", "path_length", "normal", "color_tag", "rgba")


def hits_worker(rank, world, port, width, height, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sharding_model as sharding
    from atm_raytracer_amd import synth
    from oracle_binding import Oracle
    from util import run_oracle
    cfg, tiles = synth.scene("S2", width, height, generator="Fast", max_distance=60_000.0, level=301, terrain_alpha=0.4, tilt=-4.0)
    c0, c1 = sharding.column_shard(width, rank, world)
    cfg.params.col_begin, cfg.params.col_end = c0, c1
    res = run_oracle(Oracle("det"), cfg, tiles, n_threads=2)
    slab = sharding.PlaneSlab(height, c1 - c0, torch.device("cpu"))
    for k, v in dense_planes(res).items():
        slab.planes[k].copy_(torch.from_numpy(np.ascontiguousarray(v)))
    image = sharding.ImageGather(slab, world)(dist)
    hits = {k: torch.from_numpy(np.ascontiguousarray(res[k].astype(np.int32) if k == "color_tag" else res[k])) for k in HIT_FIELDS}
    offsets, full = sharding.gather_hits(image["hit_count"], hits, world, dist)
    if rank == 0:
        queue.put((image["hit_count"].numpy(), offsets.numpy(), {k: v.numpy() for k, v in full.items()}))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_variable_length_hit_lists(oracle_det):
    """terrain_alpha < 1 (BASELINE config 5's shape of result): pixels hold several trace points, shards hold different numbers
    of them.  hit_count planes -> global offsets -> padded all-gather -> scatter must rebuild the unsharded frame's lists."""
    from atm_raytracer_amd import synth
    from util import run_oracle
    width, height, world = 40, 16, 2
    ctx = mp.get_context("spawn")
    queue = ctx.Queue()
    port = 31500 + os.getpid() % 2000
    procs = [ctx.Process(target=hits_worker, args=(r, world, port, width, height, queue)) for r in range(world)]
    for p in procs:
        p.start()
    counts, offsets, full = queue.get(timeout=180)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg, tiles = synth.scene("S2", width, height, generator="Fast", max_distance=60_000.0, level=301, terrain_alpha=0.4, tilt=-4.0)
    want = run_oracle(oracle_det, cfg, tiles)
    assert want["hit_count"].max() > 1 and np.array_equal(counts, want["hit_count"])
    assert np.array_equal(offsets, want["hit_offset"].astype(np.int64))
    for k in HIT_FIELDS:
        assert np.array_equal(full[k], want[k].astype(full[k].dtype)), k


def test_assemble_layout():
    import sharding_model as sharding
    g, h, wl = 4, 3, 5
    img = torch.arange(h * g * wl, dtype=torch.float64).reshape(h, g * wl)
    shards = torch.stack([img[:, r * wl:(r + 1) * wl] for r in range(g)])  # what all_gather_into_tensor produces
    assert torch.equal(sharding.assemble(shards), img)
    planar = torch.stack([shards, shards + 100.0], dim=1)  # [G, 3-like, H, wl] for the planar normal
    assert torch.equal(sharding.assemble(planar)[1], img + 100.0)
    assert [sharding.column_shard(4096, r, 8) for r in (0, 7)] == [(0, 512), (3584, 4096)]
    with pytest.raises(ValueError):  # unequal shards would hang the collective: refused up front
        sharding.column_shard(4096, 0, 3)


def test_plane_slab_views_alias_one_buffer():
    import sharding_model as sharding
    slab = sharding.PlaneSlab(6, 4, torch.device("cpu"))
    assert slab.nbytes == 6 * 4 * 84 and set(slab.planes) == set(PLANES)
    slab.buf.zero_()
    slab.planes["normal"][2, 5, 3] = 7.0
    slab.planes["hit_count"][0, 0] = 9
    assert slab.buf.view(torch.float64)[: 10 * 24][-1] == 7.0
    assert slab.buf[10 * 24 * 8:].view(torch.int32)[0] == 9
    pod = slab.device_planes()
    assert pod.azimuth == slab.buf.data_ptr() and pod.hit_count == slab.buf.data_ptr() + 10 * 24 * 8

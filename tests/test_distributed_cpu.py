"""The N > 1 path on the CPU: world_size 2 over gloo.  Each rank computes its pixel-column tile (here with the
oracle standing in for the HIP library — the host logic under test is the sharding, the all-gather and the
re-assembly in atm_raytracer_amd/sharding.py, exactly what bench.py runs over RCCL) and the gathered image must
equal the unsharded frame."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))
PLANES = ("azimuth", "elevation_angle", "hit_count", "first_distance", "first_lat")


def dense_planes(res):
    """Dense first-hit planes like atmrt_generate_device writes them (NaN where the pixel has no trace point)."""
    out = {"azimuth": res["azimuth"], "elevation_angle": res["elevation_angle"], "hit_count": res["hit_count"].astype(np.int32)}
    first = res["hit_offset"].astype(np.int64)
    has = res["hit_count"] > 0
    for name, src in (("first_distance", "distance"), ("first_lat", "lat")):
        plane = np.full(res["hit_count"].shape, np.nan)
        plane[has] = res[src][first[has]]
        out[name] = plane
    return out


def worker(rank, world, port, width, height, generator, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from atm_raytracer_amd import sharding, synth
    from oracle_binding import Oracle
    from util import run_oracle
    cfg, tiles = synth.scene("S2", width, height, generator=generator, max_distance=60_000.0, level=301)
    c0, c1 = sharding.column_shard(width, rank, world)
    cfg.params.col_begin, cfg.params.col_end = c0, c1
    res = run_oracle(Oracle("det"), cfg, tiles, n_threads=2)
    local = {k: torch.from_numpy(np.ascontiguousarray(v)) for k, v in dense_planes(res).items()}
    full = sharding.all_gather_planes(local, world, dist)
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
        assert got[k].shape == (height, width)
        assert np.array_equal(got[k], want[k], equal_nan=True), k


HIT_FIELDS = ("lat", "lon", "distance", "elevation", "path_length", "normal", "color_tag", "rgba")


def hits_worker(rank, world, port, width, height, queue):
    sys.path.insert(0, HERE)
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from atm_raytracer_amd import sharding, synth
    from oracle_binding import Oracle
    from util import run_oracle
    cfg, tiles = synth.scene("S2", width, height, generator="Fast", max_distance=60_000.0, level=301, terrain_alpha=0.4, tilt=-4.0)
    c0, c1 = sharding.column_shard(width, rank, world)
    cfg.params.col_begin, cfg.params.col_end = c0, c1
    res = run_oracle(Oracle("det"), cfg, tiles, n_threads=2)
    hc = torch.from_numpy(np.ascontiguousarray(res["hit_count"].astype(np.int32)))
    hits = {k: torch.from_numpy(np.ascontiguousarray(res[k].astype(np.int32) if k == "color_tag" else res[k])) for k in HIT_FIELDS}
    counts, offsets, full = sharding.all_gather_hits(hc, hits, world, dist)
    if rank == 0:
        queue.put((counts.numpy(), offsets.numpy(), {k: v.numpy() for k, v in full.items()}))
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
    from atm_raytracer_amd import sharding
    g, h, wl = 4, 3, 5
    img = torch.arange(h * g * wl, dtype=torch.float64).reshape(h, g * wl)
    shards = torch.stack([img[:, r * wl:(r + 1) * wl] for r in range(g)])  # what all_gather_into_tensor produces
    assert torch.equal(sharding.assemble(shards), img)
    planar = torch.stack([shards, shards + 100.0], dim=1)  # [G, 3-like, H, wl] for the planar normal
    assert torch.equal(sharding.assemble(planar)[1], img + 100.0)
    assert [sharding.column_shard(4096, r, 8) for r in (0, 7)] == [(0, 512), (3584, 4096)]

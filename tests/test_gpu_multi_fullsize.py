"""The multi-device route at the sizes BASELINE.json states: `atmrt_ctx_create_multi` with EIGHT sub-contexts
on the one GPU of the test box — eight host threads, eight pixel-column tiles, slabs of 352 MB at 8192x4096, the peer-copy
exchange, `k_assemble_image` over 33.5 Mpixel, the lists' blocks at their real sizes — so that the strides, the `pad256` arithmetic
and the offset scans have run at full size before an 8-GPU node runs them over RCCL.

Config 4 (8192x4096 over 5x5 tiles): the [H][W] planes of EVERY sub-context hold the oracle's bits in columns 0/1, 1023/1024 (a
tile seam), 4095/4096 (the middle seam) and 8190/8191.  Config 5 (the headline frame, terrain_alpha 0.5, 1000 objects): planes and
the complete trace-point lists equal the single-context frame (which test_gpu_baseline_configs.py pins to the oracle)."""
import numpy as np
import pytest
import torch

from atm_raytracer_amd import generators, synth
from util import bits, run_gpu, run_oracle

pytestmark = pytest.mark.gpu

DEV = torch.device("cuda", 0)
PLANES = ("azimuth", "elevation_angle", "lat", "lon", "distance", "elevation", "path_length", "normal", "hit_count")


@pytest.fixture(scope="module")
def multi8():
    ctx = generators.Context.multi([0] * 8)
    yield ctx
    ctx.close()


def same_bits(a, b):
    """Two device tensors hold the same bits (NaN payloads included: both came out of the same kernels)."""
    if a.dtype == torch.float64:
        a, b = a.view(torch.int64), b.view(torch.int64)
    return bool(torch.equal(a, b))


def check_columns_against_oracle(planes, want, c0):
    """The first-hit planes of frame columns [c0, c0 + w) against the oracle's frame of that column shard."""
    H, w = want["hit_count"].shape
    hc = want["hit_count"]
    got_hc = planes["hit_count"][:, c0:c0 + w].cpu().numpy()
    assert np.array_equal(got_hc, hc.astype(np.int32)), ("hit_count", c0)
    for k in ("azimuth", "elevation_angle"):
        assert np.array_equal(bits(planes[k][:, c0:c0 + w].cpu().numpy()), bits(want[k])), (k, c0)
    has = hc > 0
    first = want["hit_offset"].astype(np.int64)[has]
    for k in ("lat", "lon", "distance", "elevation", "path_length"):
        g = planes[k][:, c0:c0 + w].cpu().numpy()[has]
        assert np.array_equal(bits(g), bits(want[k][first])), (k, c0)
    for c in range(3):
        g = planes["normal"][c][:, c0:c0 + w].cpu().numpy()[has]
        assert np.array_equal(bits(g), bits(want["normal"][first, c])), ("normal", c, c0)
    return int(has.sum())


@pytest.mark.parametrize("generator", ["Rectilinear", "Fast"])
def test_config4_through_eight_tiles(multi8, oracle_det, generator):
    cfg, tiles = synth.scene("S4", generator=generator)
    W, H = cfg.params.width, cfg.params.height
    assert (W, H, len(tiles)) == (8192, 4096, 25)
    multi8.check(multi8.lib.atmrt_terrain_clear(multi8.handle))
    gen = generators.make_generator(generators.Params(cfg), generators.Terrain.from_tiles(tiles, multi8))
    images = [generators.image_planes(H, W, DEV) for _ in range(8)]
    steps, _ = gen.generate_image_device([pod for _, pod in images])
    tm = multi8.comm_timings()
    assert tm["world"] == 8 and tm["route"] == "peer" and tm["collectives"] == 1
    assert tm["bytes_per_rank"] >= 4096 * 1024 * 84
    assert [multi8.tile_columns(i) for i in range(8)] == [(g * 1024, (g + 1) * 1024) for g in range(8)]
    n = 0
    for c0 in (0, 1023, 4095, 8190):
        shard = synth.scene("S4", generator=generator)[0]
        shard.params.col_begin, shard.params.col_end = c0, c0 + 2
        n += check_columns_against_oracle(images[0][0], run_oracle(oracle_det, shard, tiles), c0)
    assert n > 2000
    for i in range(1, 8):  # every sub-context assembled the same image
        for k in PLANES:
            assert same_bits(images[i][0][k], images[0][0][k]), (i, k)
    print(f"config 4 {generator} through 8 tiles on one GPU: {steps} ray-steps, {n} hit pixels compared with the oracle, "
          f"slab {tm['bytes_per_rank'] / 1e6:.0f} MB per tile, exchange {tm['gather_ms']:.1f} + {tm['assemble_ms']:.1f} ms")


@pytest.mark.parametrize("generator", ["Rectilinear", "Fast"])
def test_config5_through_eight_tiles_with_lists(gpu_ctx, multi8, generator):
    tiles = synth.scene("headline", level=1)[1]
    cfg = synth.scene("headline", generator=generator, terrain_alpha=0.5)[0]
    synth.add_objects(cfg)
    W, H = cfg.params.width, cfg.params.height
    assert (W, H, len(cfg.objects)) == (4096, 2048, 1000)
    want = run_gpu(gpu_ctx, cfg, tiles)
    assert want["n_hits"] > 4_000_000 and want["hit_count"].max() > 4 and (want["color_tag"] == 1).sum() > 10_000
    multi8.check(multi8.lib.atmrt_terrain_clear(multi8.handle))
    gen = generators.make_generator(generators.Params(cfg), generators.Terrain.from_tiles(tiles, multi8))
    images = [generators.image_planes(H, W, DEV) for _ in range(8)]
    steps, _ = gen.generate_image_device([pod for _, pod in images])
    assert steps == want["ray_steps"]
    # the lists on devices 0, 3 and 7; the others take part in the collective and get nothing
    hits = gen.image_hits_device(H, W, skip=(1, 2, 4, 5, 6))
    assert multi8.comm_timings()["collectives"] == 2
    assert [h is None for h in hits] == [False, True, True, False, True, True, True, False]
    h0 = hits[0]
    assert h0["lat"].numel() == want["n_hits"]
    assert same_bits(h0["hit_offset"], torch.from_numpy(want["hit_offset"].astype(np.int64)).to(DEV))
    for k in ("lat", "lon", "distance", "elevation", "path_length", "normal", "rgba"):
        assert same_bits(h0[k], torch.from_numpy(want[k]).to(DEV)), k
    assert same_bits(h0["color_tag"], torch.from_numpy(want["color_tag"].astype(np.int32)).to(DEV))
    for i in (3, 7):
        for k, v in h0.items():
            assert same_bits(hits[i][k], v), (i, k)
    # planes: hit_count, the angles everywhere; the first trace point where there is one
    hc = torch.from_numpy(want["hit_count"].astype(np.int32)).to(DEV)
    img = images[0][0]
    assert same_bits(img["hit_count"], hc)
    for k in ("azimuth", "elevation_angle"):
        assert same_bits(img[k], torch.from_numpy(want[k]).to(DEV)), k
    first = h0["hit_offset"][hc > 0]
    for k in ("lat", "lon", "distance", "elevation", "path_length"):
        assert same_bits(img[k][hc > 0], h0[k][first]), k
    for c in range(3):
        assert same_bits(img["normal"][c][hc > 0], h0["normal"][first, c]), ("normal", c)
    for i in range(1, 8):
        for k in PLANES:
            assert same_bits(images[i][0][k], img[k]), (i, k)
    print(f"config 5 {generator} through 8 tiles: {want['n_hits']} trace points gathered on 3 of 8 devices, "
          f"tile ms {multi8.comm_timings()['tile_ms_min']:.1f} .. {multi8.comm_timings()['tile_ms_max']:.1f}")

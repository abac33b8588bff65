"""SURVEY §8(f) rank 1 — renderer::draw_image (front-to-back alpha compositing, SimpleColors / Shading palettes, fog) with
Rust's `as u8` truncation.  CPU: closed-form checks of the oracle.  GPU: k_draw_image == oracle, byte for byte."""
import ctypes as C

import numpy as np
import pytest

from atm_raytracer_amd import _abi, config, generators, synth
from util import run_gpu, run_oracle

COLORINGS = {
    "default-shading": {},
    "simple": {"coloring": {"Simple": {"water_level": 900.0}}},
    "legacy-fog": {"coloring": {"Shading": {"water_level": 700.0, "ambient_light": 0.25, "light_zenith_angle": 60.0, "light_dir": -35.0,
                                            "palette": "Legacy"}}, "fog_distance": 40_000.0},
    "simple-fog": {"coloring": {"Simple": {}}, "fog_distance": 15_000.0},
}


def scene_with_view(generator, w, h, view_extra, alpha=1.0, objects=False):
    cfg, tiles = synth.scene("S2", w, h, generator=generator, terrain_alpha=alpha, max_distance=60_000.0, tilt=-3.0)
    cfg.coloring = config._coloring(view_extra)
    if objects:
        synth.add_objects(cfg, n_cyl=30, n_bill=16, dist=(300.0, 6_000.0), spread_deg=28.0, radius=(30.0, 120.0), height=(150.0, 600.0),
                          bill_w=(150.0, 500.0), bill_h=(150.0, 500.0))
    return cfg, tiles


def test_oracle_compositing_known_answers(oracle_det):
    """One pixel, hand-made trace points: result = sum_i c_i * a_i * prod_{j<i}(1 - a_j) + sky * prod(1 - a_j), truncating to u8
    after every `add` (renderer/mod.rs:378-383, 396-411)."""
    res = {"width": 1, "height": 1, "n_hits": 2, "ray_steps": 0, "azimuth": np.zeros((1, 1)), "elevation_angle": np.zeros((1, 1)),
           "hit_count": np.array([[2]], dtype=np.uint32), "hit_offset": np.zeros((1, 1), dtype=np.uint64),
           "lat": np.zeros(2), "lon": np.zeros(2), "distance": np.array([1000.0, 2000.0]), "elevation": np.array([10.0, 20.0]),
           "path_length": np.array([1000.0, 2000.0]), "normal": np.array([[0, 0, 1.0], [0, 0, 1.0]]),
           "color_tag": np.array([1, 1], dtype=np.uint32), "rgba": np.array([[1.0, 0.0, 0.0, 0.5], [0.0, 1.0, 0.0, 1.0]])}
    col = _abi.Coloring()
    col.kind, col.palette, col.ambient_light = _abi.COLORING_SHADING, _abi.PALETTES["Improved"], 1.0  # brightness 1
    rgb = oracle_det.draw_image(res, col)[0, 0]
    # red 255 at alpha .5 -> 127; then green 255 at .5 * 1 -> (127/255, 127.5/255 -> 127, 0); sky contributes 0
    assert rgb.tolist() == [127, 127, 0]
    res["rgba"][1, 3] = 0.5
    rgb = oracle_det.draw_image(res, col)[0, 0]
    sky = [int(0.23 * 255), int(0.41 * 255), int(0.55 * 255)]
    want = [int((127 / 255 + sky[0] / 255 * 0.25) * 255), int((int((0 + 255 / 255 * 0.25) * 255) / 255 + sky[1] / 255 * 0.25) * 255),
            int((0 + sky[2] / 255 * 0.25) * 255)]
    assert rgb.tolist() == want
    res["hit_count"][0, 0] = 0
    assert oracle_det.draw_image(res, col)[0, 0].tolist() == sky
    col.has_fog, col.fog_distance = 1, 1000.0
    assert oracle_det.draw_image(res, col)[0, 0].tolist() == [160, 160, 160]  # fog colour replaces the sky (renderer/mod.rs:388-394)


def test_oracle_simple_colors_and_hsv(oracle_det, oracle_libm):
    """SimpleColors: water is blue scaled by distance; hue runs from green (low) towards red/magenta with elevation."""
    res = {"width": 3, "height": 1, "n_hits": 3, "ray_steps": 0, "azimuth": np.zeros((1, 3)), "elevation_angle": np.zeros((1, 3)),
           "hit_count": np.ones((1, 3), dtype=np.uint32), "hit_offset": np.arange(3, dtype=np.uint64).reshape(1, 3),
           "lat": np.zeros(3), "lon": np.zeros(3), "distance": np.array([0.0, 0.0, 100_000.0]), "elevation": np.array([-5.0, 0.001, 4500.0]),
           "path_length": np.zeros(3), "normal": np.tile([0, 0, 1.0], (3, 1)), "color_tag": np.zeros(3, dtype=np.uint32),
           "rgba": np.tile([0, 0, 0, 1.0], (3, 1))}
    col = _abi.Coloring()
    col.kind, col.max_distance = _abi.COLORING_SIMPLE, 200_000.0
    for o in (oracle_det, oracle_libm):
        rgb = o.draw_image(res, col)[0]
        assert rgb[0].tolist() == [0, 128, 255]                       # water at distance 0
        assert rgb[1][1] > rgb[1][0] and rgb[1][1] > rgb[1][2]        # just above water: hue ~120 (green), v = 0.9
        assert rgb[2].tolist() == [8, 8, 17]  # 4500 m at half range: hue 120 - 240 -> 240 (blue), v = 0.07, s = 0.55


@pytest.mark.parametrize("name", sorted(COLORINGS))
def test_oracle_flavours_agree_within_one_level(oracle_det, oracle_libm, name):
    cfg, tiles = scene_with_view("Fast", 64, 32, COLORINGS[name], alpha=0.6)
    imgs = []
    for o in (oracle_det, oracle_libm):
        res = run_oracle(o, cfg, tiles)
        imgs.append(o.draw_image(res, o.into_coloring(cfg.params, cfg.coloring)).astype(np.int16))
    assert np.abs(imgs[0] - imgs[1]).max() <= 1 and (imgs[0] != imgs[1]).mean() < 0.01
    assert imgs[0].std() > 5  # not a flat image


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(COLORINGS))
@pytest.mark.parametrize("generator,alpha,objects", [("Fast", 1.0, False), ("Fast", 0.5, True), ("Rectilinear", 1.0, False),
                                                       ("InterpolatingRectilinear", 0.5, False)])
def test_gpu_draw_image_matches_oracle(gpu_ctx, oracle_det, name, generator, alpha, objects):
    w, h = (96, 48) if generator != "Rectilinear" else (48, 24)
    cfg, tiles = scene_with_view(generator, w, h, COLORINGS[name], alpha, objects)
    res = run_gpu(gpu_ctx, cfg, tiles)
    col = generators.into_coloring(gpu_ctx.lib, cfg.params, cfg.coloring)
    ocol = oracle_det.into_coloring(cfg.params, cfg.coloring)
    assert bytes(col) == bytes(ocol)  # into_coloring incl. the light direction, bit for bit
    got = generators.draw_image(gpu_ctx, col, w, h)
    want = oracle_det.draw_image(res, ocol)
    assert np.array_equal(got, want)


@pytest.mark.gpu
def test_gpu_draw_image_from_dense_planes(gpu_ctx, oracle_det):
    """After atmrt_generate_device on an opaque frame the trace points exist only as dense first-hit planes."""
    import torch
    w, h = 80, 40
    cfg, tiles = scene_with_view("Fast", w, h, COLORINGS["legacy-fog"])
    res = run_gpu(gpu_ctx, cfg, tiles)
    g = generators.make_generator(generators.Params(cfg), generators.Terrain(gpu_ctx))
    dev = torch.device("cuda", 0)
    t = {k: torch.zeros((h, w), dtype=torch.float64, device=dev) for k in ("azimuth", "elevation_angle", "lat", "lon", "distance", "elevation", "path_length")}
    t["normal"] = torch.zeros((3, h, w), dtype=torch.float64, device=dev)
    t["hit_count"] = torch.zeros((h, w), dtype=torch.int32, device=dev)
    g.generate_device(_abi.DevicePlanes(**{k: v.data_ptr() for k, v in t.items()}))
    col = generators.into_coloring(gpu_ctx.lib, cfg.params, cfg.coloring)
    rgb = torch.zeros((h, w, 3), dtype=torch.uint8, device=dev)
    gpu_ctx.check(gpu_ctx.lib.atmrt_draw_image_device(gpu_ctx.handle, C.byref(col), rgb.data_ptr()))
    assert np.array_equal(rgb.cpu().numpy(), oracle_det.draw_image(res, oracle_det.into_coloring(cfg.params, cfg.coloring)))

"""SURVEY §8(f) rank 2 — the metadata file (src/generator/mod.rs:20-45 / src/viewer/mod.rs:17-29): bincode-1 layout of
AllData written byte for byte as serde derives it, checked against HAND-COMPUTED encodings, and round trips of the committed
golden frames.  The encoders are host code (no GPU needed)."""
import glob
import gzip
import os
import struct

import numpy as np
import pytest

from atm_raytracer_amd import _abi, config, metadata
from util import bits

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


def tiny_frame():
    """A 1 x 2 frame: pixel 0 holds a Terrain(0.5) point and an Rgba point, pixel 1 is sky."""
    return {"width": 2, "height": 1, "n_hits": 2, "ray_steps": 0,
            "azimuth": np.array([[10.0, 20.0]]), "elevation_angle": np.array([[1.5, -2.5]]),
            "hit_count": np.array([[2, 0]], dtype=np.uint32), "hit_offset": np.array([[0, 2]], dtype=np.uint64),
            "lat": np.array([46.0, 46.5]), "lon": np.array([8.0, 8.5]), "distance": np.array([1000.0, 2000.0]),
            "elevation": np.array([300.0, 400.0]), "path_length": np.array([1001.0, 2002.0]),
            "normal": np.array([[0.0, 0.6, 0.8], [1.0, 0.0, 0.0]]), "color_tag": np.array([0, 1], dtype=np.uint32),
            "rgba": np.array([[0.0, 0.0, 0.0, 0.5], [0.1, 0.2, 0.3, 1.0]])}


def d(*v):
    return struct.pack("<%dd" % len(v), *v)


def q(v):
    return struct.pack("<Q", v)


@pytest.mark.parametrize("prefix", [True, False])
def test_result_bytes_of_a_1x2_frame_by_hand(prefix):
    """Vec<Vec<ResultPixel>> (generators/mod.rs:13-49) under bincode 1: u64 lengths, fields in declaration order
    (elevation_angle BEFORE azimuth), u32 enum tags, Terrain(f64) / Rgba(Color{r,g,b,a})."""
    v3 = lambda x, y, z: (q(3) if prefix else b"") + d(x, y, z)
    want = q(1)                                                    # outer Vec: 1 row
    want += q(2)                                                   # row: 2 pixels
    want += d(1.5, 10.0) + q(2)                                    # pixel 0: elevation_angle, azimuth, 2 trace points
    want += d(46.0, 8.0, 1000.0, 300.0, 1001.0) + v3(0.0, 0.6, 0.8) + struct.pack("<I", 0) + d(0.5)
    want += d(46.5, 8.5, 2000.0, 400.0, 2002.0) + v3(1.0, 0.0, 0.0) + struct.pack("<I", 1) + d(0.1, 0.2, 0.3, 1.0)
    want += d(-2.5, 20.0) + q(0)                                   # pixel 1: no trace points
    got = metadata.encode_result(tiny_frame(), vector3_len_prefix=prefix).tobytes()
    assert got == want
    assert len(got) == 8 + 8 + 2 * 24 + (40 + (32 if prefix else 24) + 4 + 8) + (40 + (32 if prefix else 24) + 4 + 32)
    back, pos = metadata.decode_result(got, 0, prefix)
    assert pos == len(got)
    for k, v in tiny_frame().items():
        assert np.array_equal(np.asarray(back[k]), np.asarray(v)), k


def test_decoder_rejects_damaged_input():
    good = metadata.encode_result(tiny_frame()).tobytes()
    for bad in (good[:-1], good[:40], b"", q(1) + q(2) + good[24:], good.replace(struct.pack("<I", 1) + d(0.1), struct.pack("<I", 7) + d(0.1))):
        with pytest.raises(ValueError):
            metadata.decode_result(bad)
    ragged = q(2) + q(1) + d(0.0, 0.0) + q(0) + q(2) + (d(0.0, 0.0) + q(0)) * 2  # rows of 1 and 2 pixels
    with pytest.raises(ValueError):
        metadata.decode_result(ragged)
    with pytest.raises(ValueError):  # the other Vector3 form
        metadata.decode_result(good, 0, False)


def example_config():
    return config.Config.from_dict({
        "scene": {"terrain_folder": "dted", "terrain_alpha": 0.75,
                  "objects": [{"position": {"latitude": 46.1, "longitude": 8.2, "altitude": {"Absolute": 500.0}},
                               "shape": {"Cone": {"radius": 3.0, "height": 9.0}}, "color": {"r": 1.0, "g": 0.5, "b": 0.25}}]},
        "view": {"position": {"latitude": 46.5, "longitude": 8.5, "altitude": {"Relative": 2.0}},
                 "frame": {"direction": 90.0, "tilt": -1.0, "fov": 40.0, "max_distance": 1000.0},
                 "coloring": {"Simple": {"water_level": 3.0}}, "fog_distance": 7000.0},
        "earth_shape": {"Ellipsoid": {"a": 7.0, "b": 6.0}}, "straight_rays": True, "simulation_step": 25.0,
        "output": {"file": "o.png", "file_metadata": "m.dat", "width": 2, "height": 1, "generator": "Rectilinear",
                   "ticks": [{"Single": {"azimuth": 12.0, "size": 5, "labelled": True}}],
                   "vertical_ticks": [{"Multiple": {"bias": 0.5, "step": 2.0, "size": 3, "labelled": False}}],
                   "show_eye_level": True}})


def simple_coloring(cfg):
    col = _abi.Coloring()
    col.kind, col.water_level, col.max_distance = _abi.COLORING_SIMPLE, 3.0, cfg.params.frame.max_distance
    col.has_fog, col.fog_distance = 1, 7000.0
    return col


def test_params_bytes_by_hand():
    """`Params` (params.rs:496-505) field by field; the env segment is this package's documented stand-in."""
    cfg = example_config()
    u8, u16, u32 = (lambda v: struct.pack("<B", v)), (lambda v: struct.pack("<H", v)), (lambda v: struct.pack("<I", v))
    s = lambda t: q(len(t)) + t.encode()
    want = s("dted") + q(1)                                            # scene.terrain_folder, scene.objects
    want += d(46.1, 8.2, 500.0) + u32(0) + d(3.0, 0.0, 9.0) + d(1.0, 0.5, 0.25, 1.0)  # Coords, Shape::Frustum (a Cone: r2 = 0), Color (a defaults to 1)
    want += d(0.75)                                                    # scene.terrain_alpha
    want += d(46.5, 8.5) + u32(1) + d(2.0)                             # view.position: Altitude::Relative(2)
    want += d(90.0, -1.0, 40.0, 1000.0)                                # view.frame
    want += u32(0) + d(3.0, 1000.0)                                    # view.coloring: Simple { water_level, max_distance }
    want += u8(1) + d(7000.0)                                          # view.fog_distance: Some
    want += u32(2) + d(7.0, 6.0)                                       # model: Ellipsoid { a, b }
    env = metadata.encode_env(cfg)
    assert env.startswith(b"ATMRTENV" + u32(1) + u32(0) + d((2 * 7.0 + 6.0) / 3.0)) and env.endswith(d(530e-9))
    want += env
    want += u8(1) + d(25.0)                                            # straight_rays, simulation_step
    want += s("o.png") + u8(1) + s("m.dat") + u16(2) + u16(1)          # output.file, file_metadata: Some, width, height
    want += q(1) + u32(0) + d(12.0) + u32(5) + u8(1)                   # ticks: [Single]
    want += q(1) + u32(1) + d(0.5, 2.0) + u32(3) + u8(0)               # vertical_ticks: [Multiple]
    want += u8(1) + u8(0) + u32(2)                                     # show_eye_level, show_flat_horizon, generator: Rectilinear
    got = metadata.encode_params(cfg, [500.0], simple_coloring(cfg))
    assert got == want
    back, pos = metadata.decode_params(got)
    assert pos == len(got)
    assert back["scene"]["objects"][0]["shape"] == {"Frustum": {"r1": 3.0, "r2": 0.0, "height": 9.0}}
    assert back["view"]["coloring"] == {"Simple": {"water_level": 3.0, "max_distance": 1000.0}} and back["view"]["fog_distance"] == 7000.0
    assert back["model"] == {"Ellipsoid": {"a": 7.0, "b": 6.0}} and back["straight_rays"] is True
    assert back["output"]["ticks"] == [{"Single": {"azimuth": 12.0, "size": 5, "labelled": True}}]
    assert back["output"]["generator"] == "Rectilinear" and back["env"]["wavelength"] == 530e-9
    assert len(back["env"]["atmosphere"]["functions"]) == 7  # us_76


def test_env_converter_hook_replaces_the_unpinned_segment():
    cfg = example_config()
    marker = b"\\x01\\x02\\x03"
    got = metadata.encode_params(cfg, [500.0], simple_coloring(cfg), env_encoder=lambda c: marker)
    ref = metadata.encode_params(cfg, [500.0], simple_coloring(cfg))
    env = metadata.encode_env(cfg)
    assert got == ref.replace(env, marker) and ref.count(env) == 1


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_golden_frames_round_trip_through_the_file(path, tmp_path):
    """write_metadata -> a gzip file -> read_metadata gives back every bit of the frame (multi-hit lists, object colours)."""
    z = np.load(path)
    res = {k: z[k] for k in ("azimuth", "elevation_angle", "hit_count", "hit_offset", "lat", "lon", "distance", "elevation",
                             "path_length", "normal", "color_tag", "rgba")}
    res["height"], res["width"] = res["hit_count"].shape
    res["n_hits"], res["ray_steps"] = int(res["lat"].size), 0
    cfg = example_config()
    cfg.params.width, cfg.params.height = res["width"], res["height"]
    out = str(tmp_path / "meta.dat")
    metadata.write_metadata(out, cfg, res, simple_coloring(cfg), [500.0])
    with gzip.open(out, "rb") as f:  # a plain gzip member, like libflate's Encoder writes
        raw = f.read()
    assert raw.startswith(struct.pack("<Q", 4) + b"dted")
    back = metadata.read_metadata(out)
    assert back["params"]["output"]["width"] == res["width"] and back["params"]["output"]["height"] == res["height"]
    for k in ("azimuth", "elevation_angle", "hit_count", "hit_offset", "lat", "lon", "distance", "elevation", "path_length", "normal",
              "color_tag"):
        assert np.array_equal(bits(back["result"][k]), bits(np.asarray(res[k]))), k
    tag = res["color_tag"]
    assert np.array_equal(back["result"]["rgba"][tag == 1], res["rgba"][tag == 1])
    assert np.array_equal(back["result"]["rgba"][tag == 0][:, 3], res["rgba"][tag == 0][:, 3])  # Terrain carries its alpha only

"""DTED writer -> reader round trip (MIL-PRF-89020B records, signed-magnitude negatives) and Terrain::from_folder."""
import os

import numpy as np

from atm_raytracer_amd import synth


def test_round_trip_with_negative_posts(tmp_path, oracle_det):
    rng = np.random.default_rng(9)
    posts = rng.integers(-420, 4000, size=(121, 61)).astype(np.int16)  # n_lat != n_lon as at high latitudes
    posts[0, 0], posts[-1, -1] = -32767, 32767
    for writer in ("python", "oracle"):
        path = str(tmp_path / f"s13_w070_{writer}.dt0")
        if writer == "python":
            synth.write_dted(path, -13, -70, posts)
        else:
            oracle_det.dted_write(path, -13, -70, posts)
        lat0, lon0, back = oracle_det.dted_read(path)
        assert (lat0, lon0) == (-13, -70) and np.array_equal(back, posts)
    a = open(str(tmp_path / "s13_w070_python.dt0"), "rb").read()
    b = open(str(tmp_path / "s13_w070_oracle.dt0"), "rb").read()
    assert a[3428:] == b[3428:] and len(a) == 3428 + 61 * (12 + 2 * 121)


def test_from_folder(tmp_path, oracle_det):
    tiles = {(46, 8): synth.synth_tile(46, 8, level=101), (46, 9): synth.synth_tile(46, 9, level=101)}
    synth.write_terrain_dir(str(tmp_path / "terrain"), tiles)
    t, n = oracle_det.terrain_load_dir(str(tmp_path / "terrain"))
    assert n == 2
    assert oracle_det.get_elev(t, 46.5, 8.5) is not None and oracle_det.get_elev(t, 46.5, 9.5) is not None
    assert oracle_det.get_elev(t, 46.5, 8.995) == oracle_det.get_elev(oracle_det.terrain_new(tiles), 46.5, 8.995)
    assert oracle_det.get_elev(t, 45.5, 8.5) is None
    oracle_det.terrain_free(t)
    (tmp_path / "terrain" / "README.txt").write_text("not a tile")
    t2, n2 = oracle_det.terrain_load_dir(str(tmp_path / "terrain"))  # terrain/mod.rs:113-118: any other file is fatal
    assert n2 < 0
    empty = tmp_path / "empty"
    empty.mkdir()
    t3, n3 = oracle_det.terrain_load_dir(str(empty))  # an empty directory is fine: every lookup -> None -> 0.0
    assert n3 == 0 and oracle_det.get_elev(t3, 0.5, 0.5) is None

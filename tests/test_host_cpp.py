"""The C++ host mirror (include/atmrt_host.hpp) of the reference's generator interface: it must compile against the
C ABI (CPU check) and, on the GPU, examples/gen_host.cpp must reproduce the oracle."""
import os
import subprocess

import numpy as np
import pytest

from atm_raytracer_amd import _lib, synth
from util import run_oracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_example(out):
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "examples", "gen_host.cpp"),
                    "-o", out, "-L", _lib.CSRC, "-latmrt", f"-Wl,-rpath,{_lib.CSRC}", "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return out


def test_cpp_host_mirror_compiles_and_fails_loudly_without_gpu(tmp_path):
    exe = build_example(str(tmp_path / "gen_host"))
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    (tmp_path / "terrain").mkdir()
    r = subprocess.run([exe, str(tmp_path / "terrain"), "Fast", "8", "8", str(tmp_path / "o.bin")], capture_output=True, text=True)
    assert r.returncode == 1 and "no CPU path" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("devices", [None, "0,0,0"], ids=["one-device", "three-tiles"])
@pytest.mark.parametrize("generator", ["Fast", "Rectilinear", "InterpolatingRectilinear"])
def test_cpp_host_mirror_matches_oracle(tmp_path, oracle_det, generator, devices):
    """`devices`: the same host program with a device list — the multi-GPU path below the C ABI (three tiles on the one GPU here)."""
    exe = build_example(str(tmp_path / "gen_host"))
    tiles = synth.synth_tiles([46], [8], level=301)
    synth.write_terrain_dir(str(tmp_path / "terrain"), tiles)
    w, h = 40, 24
    out = str(tmp_path / "o.bin")
    r = subprocess.run([exe, str(tmp_path / "terrain"), generator, str(w), str(h), out] + ([devices] if devices else []), capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    assert "Detected 1 terrain files" in r.stdout and (devices is None or "3 devices" in r.stdout)
    got = np.fromfile(out, dtype=np.float64).reshape(h, w, 7)
    cfg, _ = synth.scene("S2", w, h, generator=generator, tilt=-2.0, max_distance=60_000.0)
    want = run_oracle(oracle_det, cfg, tiles)
    assert np.array_equal(got[..., 0], want["azimuth"]) and np.array_equal(got[..., 1], want["elevation_angle"])
    assert np.array_equal(got[..., 2], want["hit_count"].astype(np.float64))
    has = want["hit_count"] > 0
    first = want["hit_offset"][has].astype(np.int64)
    for i, k in enumerate(("lat", "lon", "distance", "elevation")):
        assert np.array_equal(got[..., 3 + i][has], want[k][first])
        assert np.isnan(got[..., 3 + i][~has]).all()
    if devices is None or generator != "InterpolatingRectilinear":  # tiles of the interpolating generator share lattice columns
        assert f"{want['ray_steps']} ray-steps" in r.stdout

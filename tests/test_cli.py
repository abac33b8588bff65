"""The command line (python -m atm_raytracer_amd ...) mirroring the reference's subcommands, on the GPU."""
import os
import subprocess
import sys

import numpy as np
import pytest
import yaml

from atm_raytracer_amd import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_cli(args, cwd):
    env = dict(os.environ, PYTHONPATH=ROOT)
    return subprocess.run([sys.executable, "-m", "atm_raytracer_amd"] + args, cwd=cwd, env=env, capture_output=True, text=True)


@pytest.fixture()
def workdir(tmp_path):
    synth.write_terrain_dir(str(tmp_path / "terrain"), synth.synth_tiles([46], [8], level=301))
    doc = {"scene": {"terrain_folder": "./terrain"},
           "view": {"position": {"latitude": 46.5, "longitude": 8.5, "altitude": {"Relative": 50.0}},
                    "frame": {"direction": 0.0, "fov": 60.0, "tilt": -2.0, "max_distance": 60000.0}, "fog_distance": 80000.0},
           "simulation_step": 100.0, "output": {"width": 64, "height": 32, "generator": "Fast"}}
    (tmp_path / "cfg.yaml").write_text(yaml.safe_dump(doc))
    return tmp_path


def test_cli_errors_without_gpu_or_config(tmp_path):
    r = run_cli(["output-atm", str(tmp_path / "missing.yaml")], str(tmp_path))
    assert r.returncode == 1 and "ERROR" in r.stderr


@pytest.mark.gpu
def test_cli_gen_and_diagnostics(workdir, oracle_det):
    r = run_cli(["gen", "-c", "cfg.yaml", "--output", "out.png", "--metadata", "out.npz"], str(workdir))
    assert r.returncode == 0, r.stderr
    assert "Detected 1 terrain files" in r.stdout and "Done calculating" in r.stdout
    from PIL import Image
    img = np.asarray(Image.open(workdir / "out.png"))
    assert img.shape == (32, 64, 3) and img.std() > 5
    meta = np.load(workdir / "out.npz")
    assert meta["hit_count"].shape == (32, 64) and meta["distance"].size == int(meta["hit_count"].sum())

    r = run_cli(["output-atm", "cfg.yaml", "-a", "0", "-b", "2", "-s", "1"], str(workdir))
    rows = [l.split() for l in r.stdout.strip().splitlines()]
    env = oracle_det.env()
    assert [float(x) for x in rows[1]] == [1.0, oracle_det.temperature(env, 1.0), oracle_det.pressure(env, 1.0), 0.0]

    r = run_cli(["output-ray-paths", "cfg.yaml", "-h", "10", "-a", "-0.2", "-b", "0.2", "-s", "0.2", "-c", "1000"], str(workdir))
    lines = r.stdout.strip().splitlines()
    first = lines[0].split("\t")
    assert float(first[0]) == 0.0 and [float(v) for v in first[1:4]] == [10.0, 10.0, 10.0]
    assert float(lines[-1].split("\t")[0]) >= 1000.0 and len(lines) == 21

    r = run_cli(["output-elev-profile", "cfg.yaml", "-a", "45", "-s", "500", "-c", "3000"], str(workdir))
    prof = [tuple(float(v) for v in l.split("\t")) for l in r.stdout.strip().splitlines()]
    assert [p[0] for p in prof] == [0.0, 500.0, 1000.0, 1500.0, 2000.0, 2500.0, 3000.0]
    assert all(0.0 <= p[1] <= 4000.0 for p in prof)


@pytest.mark.gpu
def test_cli_gen_writes_the_metadata_file(workdir):
    """`output.file_metadata` (generator/mod.rs:88-94): gzip(bincode(AllData)) next to the image; read back like `view` does
    (viewer/mod.rs:17-29) it must hold the frame the array dump holds, and the Params of the YAML."""
    from atm_raytracer_amd import metadata
    doc = yaml.safe_load((workdir / "cfg.yaml").read_text())
    doc["output"]["file_metadata"] = "meta.dat"
    doc["scene"]["terrain_alpha"] = 0.5
    doc["scene"]["objects"] = [{"position": {"latitude": 46.52, "longitude": 8.5, "altitude": {"Relative": 0.0}},
                                "shape": {"Cylinder": {"radius": 80.0, "height": 600.0}}, "color": {"r": 1.0, "g": 0.0, "b": 0.0, "a": 0.5}}]
    (workdir / "cfg2.yaml").write_text(yaml.safe_dump(doc))
    r = run_cli(["gen", "-c", "cfg2.yaml", "--output", "out2.png"], str(workdir))
    assert r.returncode == 0 and "Outputting metadata..." in r.stdout, r.stderr
    meta = metadata.read_metadata(str(workdir / "meta.dat"))
    r = run_cli(["gen", "-c", "cfg2.yaml", "--output", "out3.png", "--metadata", "out3.npz"], str(workdir))
    assert r.returncode == 0, r.stderr
    dump = np.load(workdir / "out3.npz")
    res = meta["result"]
    assert res["hit_count"].shape == (32, 64) and res["hit_count"].max() > 1 and (res["color_tag"] == 1).any()
    for k in ("azimuth", "elevation_angle", "hit_count", "lat", "lon", "distance", "elevation", "path_length", "normal", "color_tag"):
        assert np.array_equal(res[k], dump[k]), k
    p = meta["params"]
    assert p["scene"]["terrain_alpha"] == 0.5 and p["view"]["fog_distance"] == 80000.0 and p["output"]["file_metadata"] == "meta.dat"
    obj = p["scene"]["objects"][0]
    assert obj["shape"] == {"Frustum": {"r1": 80.0, "r2": 80.0, "height": 600.0}} and obj["position"]["elev"] > 0.0  # Altitude::abs on the terrain
    assert p["model"] == {"Spherical": {"radius": 6371000.0}} and p["simulation_step"] == 100.0

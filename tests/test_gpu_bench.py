"""bench.py as the driver runs it: `python bench.py --gpus N ...` on its own must become N ranks (it ran ONE rank
and printed "n_gpus": 1).  Rehearsed on the one GPU of the test box with two ranks sharing cuda:0 — RCCL refuses two ranks on one
device, so the exchange goes through the transport hook over gloo (ATMRT_BENCH_BACKEND=gloo); the shard assignment, the C++
image assembly and the JSON contract are the ones of the 8-GPU run."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*args, env=None, timeout=600):
    e = dict(os.environ, **(env or {}))
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True, env=e, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0]), p.stderr


@pytest.mark.parametrize("generator,extra", [("Rectilinear", []), ("Fast", ["--terrain-alpha", "0.5", "--objects", "40"]), ("Fast", [])],
                         ids=["rect-opaque", "fast-lists", "fast-opaque"])
def test_bench_gpus_2_launches_two_ranks(generator, extra):
    line, err = run_bench("--gpus", "2", "--steps", "1", "--warmup", "0", "--width", "512", "--height", "256", "--dted-level", "1",
                          "--no-cpu-baseline", "--only", "--generator", generator, *extra,
                          env={"ATMRT_BENCH_BACKEND": "gloo", "ATMRT_BENCH_CHECK_GATHER": "1"})
    assert line["n_gpus"] == 2 and line["world_size_seen"] == 2, line
    assert line["steps"] == 1 and line["value"] > 0 and line["all_gather_ms_per_step"] is not None
    assert line["all_gather_collectives_per_step"] == (1 if not extra else 2)  # the slabs (with the totals) + the lists' blocks
    assert line["config"]["parallelism"] == "pixel-column tiles x2"
    assert "gathered image check" in err and "matches the single-context frame: True" in err, err[-3000:]


def test_bench_single_gpu_line_has_the_contract_fields():
    line, _ = run_bench("--steps", "1", "--warmup", "0", "--width", "512", "--height", "256", "--dted-level", "1", "--no-cpu-baseline", "--only")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["dtype"] == "f64" and line["roofline"]["kernel"] == "k_rect_march"


def run_under_torchrun(nproc, *args, env=None, timeout=600):
    import socket
    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    e = dict(os.environ, **(env or {}))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", str(nproc), *args]
    p = subprocess.run(cmd, capture_output=True, text=True, env=e, timeout=timeout)
    assert p.returncode == 0, p.stderr[-4000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    return json.loads(lines[0]), p.stderr


@pytest.mark.parametrize("transport,route", [(None, "rccl"), ("torch", "external-device")], ids=["library-rccl", "torch-communicator"])
def test_bench_under_torchrun_one_rank_takes_the_collective_path(transport, route):
    """The driver's own launch line at N = 1: the process group exists, so the frame goes tile -> slab -> all-gather -> [H][W] —
    with the library's own ncclCommInitRank / ncclAllGather, and with the fall-back that hands the device buffers to
    torch.distributed's communicator (what bench.py switches to, on every rank, should the library's RCCL set-up fail)."""
    env = {"ATMRT_BENCH_CHECK_GATHER": "1"}
    if transport:
        env["ATMRT_BENCH_TRANSPORT"] = transport
    line, err = run_under_torchrun(1, "--steps", "1", "--warmup", "0", "--width", "512", "--height", "256", "--dted-level", "1",
                                   "--no-cpu-baseline", "--only", "--terrain-alpha", "0.5", "--objects", "30", env=env)
    assert line["n_gpus"] == 1 and line["world_size_seen"] == 1 and line["comm"]["route"] == route, line.get("comm")
    assert line["all_gather_collectives_per_step"] == 2
    assert "matches the single-context frame: True" in err, err[-3000:]

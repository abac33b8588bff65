"""The product's host-compilable object code (csrc/atmrt_objects.h: proximity filter, frustum / billboard collision, texture fetch —
what the out-of-line device functions step_object_impl, object_step_impl and close_mask_impl run) under sanitizers on the CPU, over
a randomised workload shaped like the kernels' call pattern (tests/csrc/objects_san.cpp).  ADVICE r03 asked for it: the tracer's
failure under register pressure (profiles/r04/ipra/README.md) comes and goes with register allocation, which is how latent undefined
behaviour would look too — these runs say it is not in this code: no out-of-bounds access, no undefined conversion or overflow, no
branch or index that depends on uninitialised memory.  GPU AddressSanitizer is not available on the test pool; this is the CPU run."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "csrc", "objects_san.cpp")
OUT = os.path.join(HERE, "_build")
COMMON = ["-std=c++17", "-O1", "-g", "-ffp-contract=off", "-fno-fast-math", "-fno-sanitize-recover=all"]
CLANG = "/opt/rocm/lib/llvm/bin/clang++"


def _run(compiler, flags, name, rounds):
    os.makedirs(OUT, exist_ok=True)
    exe = os.path.join(OUT, name)
    subprocess.run([compiler] + COMMON + flags + [SRC, "-o", exe, "-lm"], check=True)
    p = subprocess.run([exe, str(rounds)], capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, (p.stdout + p.stderr)[-3000:]
    assert "collisions" in p.stdout
    return p.stdout


def test_object_code_under_address_and_undefined_behaviour_sanitizers():
    out = _run("g++", ["-fsanitize=address,undefined"], "objects_san_asan_ubsan", 40)
    assert int(out.split(" proximity passes, ")[1].split(" collisions")[0]) > 10_000


@pytest.mark.skipif(not os.path.exists(CLANG) or shutil.which("ld") is None, reason="clang with the MemorySanitizer runtime is not in this image")
def test_object_code_under_memory_sanitizer():
    _run(CLANG, ["-fsanitize=memory", "-fsanitize-memory-track-origins"], "objects_san_msan", 12)

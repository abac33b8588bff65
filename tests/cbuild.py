"""Build the test-only C/C++ harnesses under tests/csrc into tests/_build (git-ignored)."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, "_build")


def _build(src, out, cmd):
    os.makedirs(OUT, exist_ok=True)
    src, out = os.path.join(HERE, "csrc", src), os.path.join(OUT, out)
    deps = [src, os.path.join(HERE, "..", "atm-raytracer_amd", "csrc", "detmath.h"),
            os.path.join(HERE, "..", "atm-raytracer_amd", "csrc", "atmrt_core.h")]
    deps.append(os.path.join(HERE, "..", "atm-raytracer_amd", "csrc", "detmath_tables.h"))
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in deps):
        subprocess.run(cmd + ["-o", out, src, "-lm"], check=True)  # -lm: fma() when the build has no -mfma
    return out


def dm_export():
    return _build("dm_export.c", "libdm_export.so", ["gcc", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math"])


def core_host():
    return _build("core_host.cpp", "libcore_host.so", ["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-ffp-contract=off", "-fno-fast-math"])


def tiff_export():
    out = os.path.join(OUT, "libtiff_export.so")
    src = os.path.join(HERE, "csrc", "tiff_export.cpp")
    hdr = os.path.join(HERE, "..", "atm-raytracer_amd", "csrc", "atmrt_tiff.h")
    os.makedirs(OUT, exist_ok=True)
    if not os.path.exists(out) or any(os.path.getmtime(d) > os.path.getmtime(out) for d in (src, hdr)):
        subprocess.run(["g++", "-std=c++17", "-O2", "-fPIC", "-shared", "-o", out, src, "-lz"], check=True)
    return out

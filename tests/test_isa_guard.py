"""The static guard against the compiler defect behind the general tracer's failure (profiles/r04/ipra/README.md part 3): hipcc 7.2
can place VGPR spill code at the head of a block AHEAD of the `s_or_b64 exec` that ends a divergent region, so that the lanes which
skipped the region never store their values.  tools/isa_prologue_spills.py finds such blocks in an assembly listing; `make check-isa`
runs it over the device code of every unit of the library, compiled with the unit's own flags.  No GPU needed."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TOOL = os.path.join(ROOT, "tools", "isa_prologue_spills.py")
FIX = os.path.join(ROOT, "tests", "golden", "asm_listings")
CSRC = os.path.join(ROOT, "atm-raytracer_amd", "csrc")


def _check(path):
    p = subprocess.run([sys.executable, TOOL, path], capture_output=True, text=True)
    return p.returncode, p.stdout


@pytest.mark.parametrize("name,label,slots", [("bad_128vgpr_LBB8_267.asm", ".LBB8_267", ("1564", "1752", "1744")),
                                              ("bad_ipra_on_LBB8_1691.asm", ".LBB8_1691", ("1400", "1352", "1344"))])
def test_checker_flags_the_two_failing_builds(name, label, slots):
    """The block of each failing build of round 4, as the compiler printed it (tests/golden/asm_listings: excerpts of `hipcc -S` output for
    this repository's own k_rect_trace): three spill stores ahead of the exec restore."""
    rc, out = _check(os.path.join(FIX, name))
    assert rc == 1 and label in out and all(f"offset:{s} " in out for s in slots), out
    assert out.strip().endswith("exec restore: 1")


def test_checker_passes_the_same_block_with_the_restore_first(tmp_path):
    """What the compiler should have emitted: the same instructions, the exec restore ahead of the stores."""
    lines = open(os.path.join(FIX, "bad_128vgpr_LBB8_267.asm")).read().splitlines()
    start = next(i for i, l in enumerate(lines) if l.startswith(".LBB8_267:"))
    restore = next(i for i in range(start, len(lines)) if "s_or_b64 exec, exec, s[2:3]" in lines[i])
    fixed = lines[:start + 1] + [lines[restore]] + lines[start + 1:restore] + lines[restore + 1:]
    f = tmp_path / "fixed.s"
    f.write_text("\n".join(fixed) + "\n")
    rc, out = _check(str(f))
    assert rc == 0 and out.strip().endswith("exec restore: 0"), out


def test_checker_ignores_whole_wave_spill_sequences_and_body_code(tmp_path):
    """`s_or_saveexec_b64 s[a:b], -1` opens a whole-wave spill of a lane-spill VGPR, not the end of a divergent region; and vector
    code behind the first non-prologue instruction of a block is the block's body, whatever restores exec later."""
    f = tmp_path / "ok.s"
    f.write_text(".LBB1_1:\n\tv_mov_b32_e32 v0, 0\n\ts_or_saveexec_b64 s[100:101], -1\n\tscratch_store_dword off, v167, off offset:1320 ; 4-byte Folded Spill\n"
                 "\ts_mov_b64 exec, s[100:101]\n.LBB1_2:\n\ts_or_b64 exec, exec, s[4:5]\n\tscratch_store_dword off, v1, off offset:8 ; 4-byte Folded Spill\n"
                 ".LBB1_3:\n\tv_add_f64 v[0:1], v[2:3], v[4:5]\n\tscratch_store_dwordx2 off, v[0:1], off offset:16\n\ts_or_b64 exec, exec, s[6:7]\n")
    rc, out = _check(str(f))
    assert rc == 0, out
    g = tmp_path / "else.s"
    g.write_text(".LBB2_1:\n\tscratch_load_dword v3, off, off offset:8 ; 4-byte Folded Reload\n\ts_or_saveexec_b64 s[0:1], s[34:35]\n\ts_xor_b64 exec, exec, s[0:1]\n")
    rc, out = _check(str(g))
    assert rc == 1 and ".LBB2_1" in out, out  # a reload ahead of the mask change that enters an else is the same defect


@pytest.mark.skipif(shutil.which("hipcc") is None and not os.path.exists("/opt/rocm/bin/hipcc"), reason="needs hipcc (cross-compiles without a GPU)")
def test_the_shipped_units_are_clean():
    """`make check-isa`: the device assembly of all nine units, every earth-model variant, with the flags the library is built with."""
    p = subprocess.run(["make", "-s", "-j8", "-C", CSRC, "check-isa"], capture_output=True, text=True, timeout=1500)
    assert p.returncode == 0 and "9 units clean" in p.stdout, (p.stdout + p.stderr)[-3000:]


OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"


@pytest.mark.skipif(not os.path.exists(OBJDUMP), reason="needs llvm-objdump")
@pytest.mark.parametrize("unit", ["atmrt_trace_linear", "atmrt_march_linear"])
def test_the_listing_is_the_code_that_ships(unit, tmp_path):
    """The guard reads `hipcc -S` output, the library links `hipcc -c` output: same flags, and the same instructions — the device code
    object inside the shipped .o, disassembled, holds exactly as many spill stores, reloads, lane writes, calls, exec restores and FMAs
    as the listing check-isa looked at."""
    obj, lst = os.path.join(CSRC, unit + ".o"), os.path.join(CSRC, "isa", unit + ".s")
    if not (os.path.exists(obj) and os.path.exists(lst)):
        pytest.skip("library or listings not built (make -C atm-raytracer_amd/csrc; make check-isa)")
    if os.path.getmtime(lst) < os.path.getmtime(os.path.join(CSRC, "atmrt_march_impl.h")):
        pytest.skip("listing older than the sources: test_the_shipped_units_are_clean rebuilds it")
    shutil.copy(obj, tmp_path / "u.o")
    subprocess.run([OBJDUMP, "--offloading", "u.o"], cwd=tmp_path, check=True, capture_output=True)  # writes u.o.0.hipv4-amdgcn-...
    co = [f for f in os.listdir(tmp_path) if "amdgcn" in f]
    assert len(co) == 1, os.listdir(tmp_path)
    dis = subprocess.run([OBJDUMP, "-d", co[0]], cwd=tmp_path, check=True, capture_output=True, text=True).stdout
    listing = open(lst).read()
    for op in ("scratch_store_dword", "scratch_load_dword", "s_swappc_b64", "v_writelane_b32", "s_or_b64 exec, exec", "v_fma_f64", "v_rcp_f64"):
        in_obj = sum(1 for l in dis.splitlines() if op in l)
        in_lst = sum(1 for l in listing.splitlines() if l.lstrip().startswith(op))
        assert in_obj == in_lst and in_obj > 0, (unit, op, in_obj, in_lst)

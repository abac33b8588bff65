#!/usr/bin/env python3
"""Static check of an AMDGPU assembly listing (hipcc -S --cuda-device-only) for the code-generation defect behind the general
tracer's failure (profiles/r04/ipra/README.md part 3): VGPR spill code placed in a block's prologue AHEAD of the instruction that
restores the exec mask.

A block that ends a divergent region begins with its prologue: SGPR spills into VGPR lanes (v_writelane, exec-independent) and the
exec restore `s_or_b64 exec, exec, s[a:b]`.  Spill stores, reloads and split copies of VGPRs that the register allocator wants "at
the top of the block" belong BEHIND that restore: ahead of it they run under the narrowed mask of the region that just ended, so the
lanes that skipped the region never store (or reload) their values.  hipcc 7.2 puts them ahead of it when an IMPLICIT_DEF of a
lane-spill VGPR sits among the prologue's v_writelanes (the scan for the end of the prologue stops there).

    python3 tools/isa_prologue_spills.py listing.s [function-substring]

prints every such block; exit code 1 if there is one."""
import re
import sys

LABEL = re.compile(r"^(\.LBB\d+_\d+|_Z\w+):")
EXEC_RESTORE = re.compile(r"^\s*(s_or_b64\s+exec,\s*exec,\s*s\[\d+:\d+\]|s_or_saveexec_b64\s+s\[\d+:\d+\],\s*s\[)")  # end of an if / a loop; entry of an else (not the `s_or_saveexec_b64 s[a:b], -1` of a whole-wave spill)
HARMLESS = re.compile(r"^\s*(v_writelane_b32|s_nop|s_waitcnt|v_readlane_b32|s_mov_b32|s_mov_b64)\b")  # exec-independent (scalar / lane moves)
VECTOR = re.compile(r"^\s*(scratch_store|scratch_load|buffer_store|buffer_load|v_mov_b32|v_mov_b64|v_accvgpr)")


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    func, bad, lines = "", [], open(path).read().splitlines()
    i = 0
    while i < len(lines):
        m = LABEL.match(lines[i])
        if not m:
            i += 1
            continue
        if m.group(1).startswith("_Z"):
            func = m.group(1)
            i += 1
            continue
        label, j, vec = m.group(1), i + 1, []
        while j < len(lines):
            t = lines[j]
            s = t.strip()
            if not s or s.startswith(";"):
                j += 1
                continue
            if EXEC_RESTORE.match(t):
                if vec and want in func:
                    bad.append((func, label, i + 1, vec))
                break
            if VECTOR.match(t):
                vec.append((j + 1, s))
            elif not HARMLESS.match(t):
                break  # the body has begun: whatever restores exec later is not this block's prologue
            j += 1
        i += 1
    for func, label, ln, vec in bad:
        print(f"{func[:70]} {label} (line {ln}): {len(vec)} vector memory / move instruction(s) ahead of the block's exec restore")
        for l, s in vec[:6]:
            print(f"    {l}: {s}")
    print(f"blocks with vector spill code ahead of their exec restore: {len(bad)}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

#!/usr/bin/env python3
"""Static check of the register contract across device-function calls in an AMDGPU assembly listing (hipcc -S --cuda-device-only).

For every `s_swappc_b64` in every function: the registers LIVE across the call in the caller (backward data-flow over the
function's control-flow graph, 32-bit register granularity) against the registers the callee WRITES (transitively over its own
calls, minus the ones its prologue saves to scratch and its epilogue restores).  A register in both sets is a value the caller
expects to survive a call that overwrites it.

Written for VERDICT r02 item 3 (the general tracer lost its trace points with interprocedural register allocation enabled):
    python3 tools/isa_call_liveness.py /tmp/isa/trace_ipra.s [function-substring]
"""
import re
import sys
from collections import defaultdict

REG = re.compile(r"\b([vsa])(\d+)\b|\b([vsa])\[(\d+):(\d+)\]|\b(vcc|exec|vcc_lo|vcc_hi|exec_lo|exec_hi|scc|m0|flat_scratch)\b")
NO_DEF = ("s_cmp", "s_bitcmp", "v_cmpx", "s_cbranch", "s_branch", "s_waitcnt", "s_nop", "s_endpgm", "s_barrier", "s_setprio", "s_sleep",
          "global_store", "scratch_store", "flat_store", "buffer_store", "ds_write", "ds_store", "s_setpc", "s_sethalt", "s_trap",
          "s_setreg", "s_dcache", "s_icache", "buffer_wbl2", "buffer_inv", "s_sendmsg", "s_code_end", "s_inst_prefetch", "global_atomic_add_u64",
          "global_atomic_add_u32", "ds_add_u32", "ds_add_u64", "global_atomic_umax", "global_atomic_smax", "s_ttracedata", "s_wait")
TWO_DEFS = ("v_div_scale", "v_add_co", "v_sub_co", "v_subrev_co", "v_addc_co", "v_subb_co", "v_subbrev_co", "v_mad_u64_u32", "v_mad_i64_i32",
            "s_swappc")
ALSO_USES_DST = ("v_fmac", "v_mac", "v_writelane", "v_cndmask", "v_movrel", "s_cmov", "v_dot", "v_pk_fmac", "s_bitset", "v_mfma", "v_accvgpr",
                 "v_fmaak", "v_fmamk")
SETS_SCC = ("s_add", "s_sub", "s_and", "s_or", "s_xor", "s_andn2", "s_orn2", "s_nand", "s_nor", "s_xnor", "s_lshl", "s_lshr", "s_ashr", "s_bfe",
            "s_cmp", "s_bitcmp", "s_min", "s_max", "s_abs", "s_not", "s_wqm", "s_quadmask", "s_bcnt", "s_addc", "s_subb", "s_absdiff", "s_mul_hi")
SAVEEXEC = ("saveexec",)


def regs_of(tok):
    out = []
    for m in REG.finditer(tok):
        if m.group(1):
            out.append((m.group(1), int(m.group(2))))
        elif m.group(3):
            out += [(m.group(3), i) for i in range(int(m.group(4)), int(m.group(5)) + 1)]
        else:
            n = m.group(6)
            out += {"vcc": [("vcc", 0), ("vcc", 1)], "exec": [("exec", 0), ("exec", 1)], "vcc_lo": [("vcc", 0)], "vcc_hi": [("vcc", 1)],
                    "exec_lo": [("exec", 0)], "exec_hi": [("exec", 1)], "scc": [("scc", 0)], "m0": [("m0", 0)],
                    "flat_scratch": [("fs", 0), ("fs", 1)]}[n]
    return out


def split_operands(rest):
    rest = rest.split(";")[0]
    out, depth, cur = [], 0, ""
    for ch in rest:
        if ch == "[":
            depth += 1
        if ch == "]":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


class Ins:
    __slots__ = ("op", "defs", "uses", "target", "text", "callee")

    def __init__(self, text):
        self.text = text
        parts = text.split(None, 1)
        self.op = parts[0]
        ops = split_operands(parts[1]) if len(parts) > 1 else []
        self.target = None
        self.callee = None
        defs, uses = [], []
        op = self.op
        if op.startswith(("s_cbranch", "s_branch")):
            self.target = ops[0] if ops else None
            if "vcc" in op:
                uses += [("vcc", 0), ("vcc", 1)]
            if "exec" in op:
                uses += [("exec", 0), ("exec", 1)]
            if "scc" in op:
                uses += [("scc", 0)]
        elif op.startswith(NO_DEF):
            for o in ops:
                uses += regs_of(o)
            if op.startswith(("s_cmp", "s_bitcmp")):
                defs.append(("scc", 0))
            if op.startswith("v_cmpx"):
                defs += [("exec", 0), ("exec", 1)]
        else:
            nd = 2 if op.startswith(TWO_DEFS) else 1
            for i, o in enumerate(ops):
                (defs if i < nd else uses).extend(regs_of(o))
            if op.startswith(ALSO_USES_DST) or "_sdwa" in op or "_dpp" in op or "row_" in text or "quad_perm" in text:
                uses += defs
            if op.startswith("v_cmp") and ops and not ops[0].startswith(("s", "vcc")):  # e32 form: implicit vcc
                uses += defs
                defs = [("vcc", 0), ("vcc", 1)]
            if op.startswith(SETS_SCC):
                defs.append(("scc", 0))
            if any(k in op for k in SAVEEXEC):
                defs += [("exec", 0), ("exec", 1)]
                uses += [("exec", 0), ("exec", 1)]
            if op.startswith(("v_addc_co", "v_subb_co", "v_subbrev_co", "v_cndmask_b32_e32", "v_div_fmas")):
                uses += [("vcc", 0), ("vcc", 1)]
            if op.startswith(("s_addc", "s_subb", "s_cselect", "s_cmov")):
                uses.append(("scc", 0))
            if op.startswith(("v_", "global_", "scratch_", "flat_", "ds_", "buffer_")):
                uses += [("exec", 0), ("exec", 1)]
            if op.startswith(("scratch_load", "scratch_store")) or op.startswith("scratch_"):
                uses += [("s", 32), ("s", 33)] if False else []
        self.defs, self.uses = set(defs), set(uses)


def parse(path):
    funcs, cur, name = {}, None, None
    for line in open(path):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            name = m.group(1)
            cur = funcs[name] = []
            continue
        if cur is None:
            continue
        if re.match(r"^\s*\.(end_amdhsa_kernel|size|section|Lfunc_end)", line) or line.startswith(".Lfunc_end"):
            if line.startswith(".Lfunc_end") or ".size" in line:
                cur = None
            continue
        m = re.match(r"^(\.LBB\d+_\d+):", line)
        if m:
            cur.append(("label", m.group(1)))
            continue
        t = line.strip()
        if not t or t.startswith((";", ".")):
            continue
        cur.append(("ins", Ins(t)))
    return funcs


def find_callees(body, names):
    """s_getpc / s_add_u32 ... @rel32 sequences name the callee in a symbol reference shortly before each s_swappc."""
    sites = []
    last = None
    for kind, x in body:
        if kind != "ins":
            continue
        m = re.search(r"(_Z\w+)@(?:rel32|gotpcrel32)", x.text)
        if m:
            last = m.group(1)
        if x.op.startswith("s_swappc"):
            x.callee = last
            sites.append(x)
    return sites


def written(funcs, name, memo):
    if name in memo:
        return memo[name]
    memo[name] = set()
    body = funcs.get(name, [])
    w = set()
    saved, restored = set(), set()
    for kind, x in body:
        if kind != "ins":
            continue
        if x.op.startswith("scratch_store") and "s33" in x.text or x.op.startswith("scratch_store") and "s32" in x.text:
            saved |= {r for r in regs_of(x.text.split(None, 1)[1].split(",")[1]) if r[0] == "v"} if "," in x.text else set()
        if x.op.startswith("scratch_load"):
            restored |= {r for r in x.defs if r[0] == "v"}
        w |= x.defs
        if x.op.startswith("s_swappc") and x.callee:
            w |= written(funcs, x.callee, memo)
    memo[name] = (w, saved & restored)
    return memo[name]


def liveness(body):
    # basic blocks
    blocks, cur, label = [], [], "entry"
    for kind, x in body:
        if kind == "label":
            blocks.append((label, cur))
            label, cur = x, []
        else:
            cur.append(x)
            if x.op.startswith(("s_branch", "s_endpgm", "s_setpc")):
                blocks.append((label, cur))
                label, cur = f"{label}.after{len(blocks)}", []
    blocks.append((label, cur))
    index = {lab: i for i, (lab, _) in enumerate(blocks)}
    succ = []
    for i, (lab, ins) in enumerate(blocks):
        s = set()
        fall = True
        for x in ins:
            if x.target and x.target in index:
                s.add(index[x.target])
            if x.op.startswith(("s_branch", "s_endpgm", "s_setpc")):
                fall = False
        if fall and i + 1 < len(blocks):
            s.add(i + 1)
        succ.append(s)
    use, deff = [], []
    for lab, ins in blocks:
        u, d = set(), set()
        for x in ins:
            u |= x.uses - d
            d |= x.defs
        use.append(u)
        deff.append(d)
    live_in = [set() for _ in blocks]
    live_out = [set() for _ in blocks]
    changed = True
    while changed:
        changed = False
        for i in reversed(range(len(blocks))):
            out = set()
            for s in succ[i]:
                out |= live_in[s]
            inn = use[i] | (out - deff[i])
            if out != live_out[i] or inn != live_in[i]:
                live_out[i], live_in[i] = out, inn
                changed = True
    return blocks, live_out


def fmt(regs):
    by = defaultdict(list)
    for k, i in sorted(regs):
        by[k].append(i)
    return " ".join(f"{k}{v}" for k, v in by.items())


def main():
    path = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    funcs = parse(path)
    for name, body in funcs.items():
        find_callees(body, funcs)
    memo = {}
    total_bad = 0
    for name, body in funcs.items():
        if want not in name:
            continue
        sites = [x for k, x in body if k == "ins" and x.op.startswith("s_swappc")]
        if not sites:
            continue
        blocks, live_out = liveness(body)
        for bi, (lab, ins) in enumerate(blocks):
            for j, x in enumerate(ins):
                if not x.op.startswith("s_swappc"):
                    continue
                live = set(live_out[bi])
                for y in reversed(ins[j + 1:]):
                    live = (live - y.defs) | y.uses
                w, preserved = written(funcs, x.callee, memo) if x.callee else (set(), set())
                clash = {r for r in (live & (w - preserved)) if r[0] in ("v", "s", "a")}
                clash -= {("s", 30), ("s", 31)}
                total_bad += bool(clash)
                print(f"{name[:60]} {lab}: call {x.callee[:40] if x.callee else '?'}: live across {len(live)} regs, callee writes {len(w)} "
                      f"(restores {len(preserved)}); LIVE AND WRITTEN: {fmt(clash) if clash else 'none'}")
    print(f"call sites with a live register the callee writes: {total_bad}")


if __name__ == "__main__":
    main()

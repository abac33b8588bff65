#!/usr/bin/env python3
"""Turns the rocprofv3 outputs of tools/collect_profiles.sh (one generator per pass) into the summaries kept under profiles/:

    python3 tools/summarize_profiles.py gpurun_out/final profiles/r01 v4

  kernel_stats_{rect,fast,interp}_<tag>.csv   rocprofv3's own kernel_stats.csv of each generator's bench run
  pmc_hbm_summary_<tag>.json     FETCH_SIZE / WRITE_SIZE per launch per kernel (KB as reported; bytes raw and with FETCH doubled,
                                 the gfx950 correction of MI355X_MICROARCH.md) — also written to profiles/pmc_hbm_latest.json
  sq_counters_<tag>.json         per-launch SQ counter sums, VALU-busy fraction (SQ_ACTIVE_INST_VALU x 4 / 1024 SIMDs / kernel
                                 cycles, kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs), active lanes, VALU lane-instructions per ray-step
  bench_*_<tag>.json             the bench lines printed by the same runs
"""
import csv
import glob
import json
import os
import re
import shutil
import sys
from collections import defaultdict

N_XCD, N_SIMD = 8, 1024
KERNEL = re.compile(r"(atmrt::\w+(?:<[^(]*>)?)\(")  # "void atmrt::k_x<8, 0>(args...)" -> "atmrt::k_x<8, 0>"


def counters(dirname):
    """{kernel: {counter: mean per launch}}, launches per kernel."""
    files = glob.glob(os.path.join(dirname + "_*", "**", "*counter_collection.csv"), recursive=True)
    per = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))  # kernel -> counter -> dispatch -> value
    for f in files:
        for row in csv.DictReader(open(f, newline="")):
            m = KERNEL.search(row["Kernel_Name"])
            if m:
                k = m.group(1)
                per[k][row["Counter_Name"]][row["Dispatch_Id"]] += float(row["Counter_Value"])
    out, launches = {}, {}
    for k, cs in per.items():
        out[k] = {c: sum(d.values()) / len(d) for c, d in cs.items()}
        launches[k] = max(len(d) for d in cs.values())
    return out, launches


def bench_line(path):
    for line in open(path):
        if line.startswith("{"):
            return json.loads(line)
    return None


def meta(tag):
    """What bench.py checks before quoting these counters: the hash of the library sources they were collected from."""
    import datetime
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
    from atm_raytracer_amd import _lib
    return {"source_hash": _lib.source_hash(), "collected": datetime.date.today().isoformat(), "tag": tag,
            "command": "bench.py --steps 3 --warmup 1 --no-cpu-baseline --only --generator G (tools/collect_profiles.sh)"}


def main():
    src, dst, tag = sys.argv[1], sys.argv[2], sys.argv[3]
    os.makedirs(dst, exist_ok=True)
    copies = [("bench_default.json", f"bench_default_{tag}.json")]
    for gen, short in (("Rectilinear", "rect"), ("Fast", "fast"), ("InterpolatingRectilinear", "interp")):
        for f in glob.glob(os.path.join(src, f"trace_{gen}", "**", "*kernel_stats.csv"), recursive=True)[:1]:
            shutil.copy(f, os.path.join(dst, f"kernel_stats_{short}_{tag}.csv"))
        copies.append((f"bench_trace_{gen}.json", f"bench_under_rocprof_{short}_{tag}.json"))
    for name, to in copies:
        if os.path.exists(os.path.join(src, name)):
            shutil.copy(os.path.join(src, name), os.path.join(dst, to))

    fetch, nf = counters(os.path.join(src, "fetch"))
    write, nw = counters(os.path.join(src, "write"))
    hbm = {}
    gen_frames = {}
    for gen in ("Rectilinear", "Fast"):
        q = os.path.join(src, f"bench_fetch_{gen}.json")
        if os.path.exists(q) and bench_line(q):
            gen_frames[gen] = bench_line(q)["steps"] + bench_line(q)["warmup"]
    for k in sorted(set(fetch) | set(write)):
        f_kb, w_kb = fetch.get(k, {}).get("FETCH_SIZE", 0.0), write.get(k, {}).get("WRITE_SIZE", 0.0)
        hbm[k] = {"FETCH_SIZE_KB_per_launch": f_kb, "WRITE_SIZE_KB_per_launch": w_kb,
                  "hbm_bytes_per_launch_raw": (f_kb + w_kb) * 1024.0, "hbm_bytes_per_launch_fetch_x2": (2.0 * f_kb + w_kb) * 1024.0,
                  "launches_FETCH_SIZE": nf.get(k, 0), "launches_WRITE_SIZE": nw.get(k, 0)}
        frames = gen_frames.get("Rectilinear" if "k_rect" in k else "Fast")
        if frames:  # kernels launched in segments: traffic of one frame
            per_frame = nf.get(k, 0) / frames
            hbm[k]["launches_per_frame"] = per_frame
            hbm[k]["hbm_bytes_per_frame_fetch_x2"] = hbm[k]["hbm_bytes_per_launch_fetch_x2"] * per_frame
    if hbm:
        hbm["_meta"] = meta(tag)
        for p in (os.path.join(dst, f"pmc_hbm_summary_{tag}.json"), os.path.join(os.path.dirname(dst.rstrip("/")), "pmc_hbm_latest.json")):
            json.dump(hbm, open(p, "w"), indent=1, sort_keys=True)

    sq = defaultdict(dict)
    sq_launches = {}
    for sub in ("sq1", "sq2", "mix1", "mix2"):
        c, nl = counters(os.path.join(src, sub))
        for k, v in c.items():
            sq[k].update(v)
            sq_launches[k] = nl.get(k, 0)
    steps, frames = {}, {}
    for gen, frag in (("Rectilinear", "k_rect_march"), ("Fast", "k_fast_intersect")):
        p = os.path.join(src, f"bench_sq1_{gen}.json")
        if os.path.exists(p) and bench_line(p):
            steps[frag] = bench_line(p)["ray_steps_per_frame"]
            frames[frag] = bench_line(p)["steps"] + bench_line(p)["warmup"]
    # mean launch duration of every kernel from rocprofv3's own kernel_stats.csv of the trace passes (AverageNs)
    avg_ns = {}
    for gen in ("Rectilinear", "Fast", "InterpolatingRectilinear"):
        for f in glob.glob(os.path.join(src, f"trace_{gen}", "**", "*kernel_stats.csv"), recursive=True)[:1]:
            for row in csv.DictReader(open(f, newline="")):
                m = KERNEL.search(row["Name"])
                if m and m.group(1) not in avg_ns:
                    avg_ns[m.group(1)] = float(row["AverageNs"])
    for k, v in sq.items():
        if "GRBM_GUI_ACTIVE" in v and "SQ_ACTIVE_INST_VALU" in v:
            v["kernel_cycles"] = v["GRBM_GUI_ACTIVE"] / N_XCD
            v["valu_busy_frac"] = v["SQ_ACTIVE_INST_VALU"] * 4.0 / N_SIMD / v["kernel_cycles"]
            if k in avg_ns:  # the clock the kernel actually ran at: GRBM_GUI_ACTIVE / 8 XCDs / its mean duration
                v["kernel_ms_rocprof"] = avg_ns[k] / 1e6
                v["effective_clock_ghz"] = v["kernel_cycles"] / avg_ns[k]
        if "SQ_THREAD_CYCLES_VALU" in v and v.get("SQ_ACTIVE_INST_VALU"):
            v["lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU"] / (64.0 * v["SQ_ACTIVE_INST_VALU"])
        for frag, n in steps.items():
            if frag in k and "SQ_INSTS_VALU" in v and "lane_utilisation" in v and ("march<0" in k or "intersect<8, 0" in k):
                per_frame = sq_launches.get(k, 0) / frames[frag] if frames.get(frag) else 1.0  # a kernel launched in segments
                v["launches_per_frame"] = per_frame
                v["valu_lane_instructions_per_ray_step"] = v["SQ_INSTS_VALU"] * per_frame * 64.0 * v["lane_utilisation"] / n
                if "SQ_INSTS_VALU_FMA_F64" in v:
                    # The instruction mix (wave-instructions per launch -> lane-instructions per ray-step) and the FP64 FLOP count it
                    # implies: an FMA is 2 flops, an add or a mul 1, a transcendental (v_rcp_f64, v_rsq_f64 ...) 1.  With
                    # -ffp-contract=off the formulas of the reference compile to separate mul and add; FMAs come from detmath's
                    # explicit fma() (polynomials, division / square-root refinement).
                    scale = per_frame * 64.0 * v["lane_utilisation"] / n
                    mix = {k2: v.get("SQ_INSTS_VALU_" + k2, 0.0) * scale for k2 in
                           ("ADD_F64", "MUL_F64", "FMA_F64", "TRANS_F64", "INT32", "INT64", "CVT", "ADD_F32", "MUL_F32", "FMA_F32", "TRANS_F32")}
                    mix["other_moves_compares_selects"] = v["valu_lane_instructions_per_ray_step"] - sum(mix.values())
                    v["lane_instructions_per_ray_step_by_kind"] = mix
                    v["fp64_flops_per_ray_step"] = mix["ADD_F64"] + mix["MUL_F64"] + 2.0 * mix["FMA_F64"] + mix["TRANS_F64"]
                    v["fp64_arith_lane_instructions_per_ray_step"] = mix["ADD_F64"] + mix["MUL_F64"] + mix["FMA_F64"] + mix["TRANS_F64"]
                    if "SQ_INSTS_VALU_FLOPS_FP64" in v:  # the hardware's own flop counter, same normalisation (cross-check)
                        v["fp64_flops_per_ray_step_hw_counter"] = (v["SQ_INSTS_VALU_FLOPS_FP64"] + v.get("SQ_INSTS_VALU_FLOPS_FP64_TRANS", 0.0)) * scale
    if sq:
        sq["_meta"] = meta(tag)
        for p in (os.path.join(dst, f"sq_counters_{tag}.json"), os.path.join(os.path.dirname(dst.rstrip("/")), "sq_counters_latest.json")):
            json.dump(sq, open(p, "w"), indent=1, sort_keys=True)
    print(f"kernels: hbm {len(hbm)}, sq {len(sq)}; written to {dst} with tag {tag}")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Summary of tools/collect_shard_profiles.sh:  python3 tools/summarize_shard_profiles.py gpurun_out/r03_shards profiles/r03 [G]
writes kernel_stats_shards<G>.csv (the rocprofv3 --stats table) and shard_march_profile.json (per kernel and launch: average
duration, HBM bytes written / fetched as the PMC counters report them, VALU-busy fraction, lane utilisation, wavefronts)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atm_raytracer_amd import _lib  # noqa: E402

src, dst = sys.argv[1], sys.argv[2]
G = sys.argv[3] if len(sys.argv) > 3 else "8"
N_XCD, N_SIMD = 8, 1024
shutil.copy(os.path.join(src, "trace", "t_kernel_stats.csv"), os.path.join(dst, f"kernel_stats_shards{G}.csv"))
out = collections.defaultdict(dict)
for r in csv.DictReader(open(os.path.join(src, "trace", "t_kernel_stats.csv"))):
    if "atmrt::" in r["Name"]:
        k = r["Name"].split("(")[0].replace("void ", "")
        out[k].update(launches=int(r["Calls"]), average_ms=float(r["AverageNs"]) / 1e6, percentage=float(r["Percentage"]))
for name in ("write", "fetch", "sq"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(os.path.join(src, name, "p_counter_collection.csv"))):
        agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        if k in out:
            for c, x in v.items():
                out[k][c + "_per_launch"] = sum(x) / len(x)
for k, v in out.items():
    if "WRITE_SIZE_per_launch" in v:
        v["hbm_written_MB_per_launch"] = v.pop("WRITE_SIZE_per_launch") * 1024 / 1e6   # counter unit: KB
        v["hbm_fetched_MB_per_launch_raw"] = v.pop("FETCH_SIZE_per_launch") * 1024 / 1e6
        v["hbm_fetched_MB_per_launch_x2"] = 2 * v["hbm_fetched_MB_per_launch_raw"]        # gfx950 note, MI355X_MICROARCH.md
    if "GRBM_GUI_ACTIVE_per_launch" in v:
        cyc = v["GRBM_GUI_ACTIVE_per_launch"] / N_XCD
        v["valu_busy_frac"] = v["SQ_ACTIVE_INST_VALU_per_launch"] * 4.0 / N_SIMD / cyc
        v["lane_utilisation"] = v["SQ_THREAD_CYCLES_VALU_per_launch"] / (64.0 * v["SQ_ACTIVE_INST_VALU_per_launch"])
        v["mean_occupancy_waves_per_simd"] = v["SQ_WAVE_CYCLES_per_launch"] / N_SIMD / (v["SQ_BUSY_CYCLES_per_launch"] / N_SIMD * 4) if 0 else None
        v.pop("mean_occupancy_waves_per_simd")
bal = json.loads(open(os.path.join(src, "balance_plain.json")).read().strip().splitlines()[-1])
out["_shards"] = bal
out["_meta"] = {"source_hash": _lib.source_hash(), "command": f"tools/collect_shard_profiles.sh (tools/measure_shard_balance.py {G}: every shard of a {G}-GPU run of the headline, 6 frames each)",
                "algorithmic_store_MB_per_shard": 84 * 4096 * 2048 / int(G) / 1e6}
json.dump(out, open(os.path.join(dst, f"shard_march_profile_{G}.json"), "w"), indent=1, sort_keys=True)
for k, v in out.items():
    if not k.startswith("_"):
        print(k, {a: (round(b, 3) if isinstance(b, float) else b) for a, b in v.items() if not a.endswith("_per_launch") or "MB" in a})

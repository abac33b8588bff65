#!/bin/bash
# BASELINE.json configs 3, 4 and 5 on one GPU (the headline, config 2's metric settings at 4096x2048, is bench.py's default):
#   gpurun --timeout 1100 -- 'bash tools/measure_baseline_configs.sh gpurun_out/cfgs'
# then  python3 tools/measure_baseline_configs.sh --collect gpurun_out/cfgs profiles/r01/baseline_configs_3_4_5.json  (see below)
set -e
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/${1:-gpurun_out/cfgs}
mkdir -p "$OUT"
B="python3 $REPO/bench.py --no-cpu-baseline --only --steps 3 --warmup 2"
$B --scene S3 --step 50 --generator Rectilinear > "$OUT/C3_Rectilinear.json" 2> "$OUT/C3_Rectilinear.err"
$B --scene S3 --step 50 --generator Fast > "$OUT/C3_Fast.json" 2> "$OUT/C3_Fast.err"
echo "[configs] C3 done"
$B --scene S4 --width 8192 --height 4096 --dted-level 1 --generator Rectilinear > "$OUT/C4_Rectilinear.json" 2> "$OUT/C4_Rectilinear.err"
$B --scene S4 --width 8192 --height 4096 --dted-level 1 --generator Fast > "$OUT/C4_Fast.json" 2> "$OUT/C4_Fast.err"
echo "[configs] C4 done"
$B --objects 1000 --terrain-alpha 0.5 --generator Rectilinear > "$OUT/C5_Rectilinear.json" 2> "$OUT/C5_Rectilinear.err"
$B --objects 1000 --terrain-alpha 0.5 --generator Fast > "$OUT/C5_Fast.json" 2> "$OUT/C5_Fast.err"
$B --objects 1000 --terrain-alpha 0.5 --generator InterpolatingRectilinear > "$OUT/C5_Interpolating.json" 2> "$OUT/C5_Interpolating.err"
echo "[configs] C5 done"

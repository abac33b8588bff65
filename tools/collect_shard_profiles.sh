#!/bin/bash
# rocprofv3 evidence for the time-sliced march of a column shard (the kernels a multi-GPU rank runs; bench.py at N=1 runs the plain
# k_rect_march):  gpurun --timeout 900 -- 'bash tools/collect_shard_profiles.sh gpurun_out/shards 8'
# Passes: kernel trace + stats, WRITE_SIZE, FETCH_SIZE, SQ busy counters — each its own run of tools/measure_shard_balance.py G
# (every shard of a G-GPU run of the headline, 6 frames each).  Summarised by tools/summarize_shard_profiles.py.
set -e
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/${1:-gpurun_out/shards}
G=${2:-8}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
RUN="$REPO/tools/measure_shard_balance.py $G"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -o t -- python3 $RUN > "$OUT/balance_trace.json" 2> "$OUT/trace.err"
echo "[shards] kernel trace done"
rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/write" -o p -- python3 $RUN > "$OUT/balance_write.json" 2> "$OUT/write.err"
rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch" -o p -- python3 $RUN > "$OUT/balance_fetch.json" 2> "$OUT/fetch.err"
echo "[shards] WRITE_SIZE / FETCH_SIZE done"
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE -d "$OUT/sq" -o p -- python3 $RUN > "$OUT/balance_sq.json" 2> "$OUT/sq.err"
echo "[shards] SQ pass done"
python3 $RUN > "$OUT/balance_plain.json" 2> "$OUT/plain.err"
find "$OUT" -name "*.csv" -size +20M -delete

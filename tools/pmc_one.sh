#!/bin/bash
# SQ counters of an arbitrary bench.py invocation:  bash tools/pmc_one.sh OUTDIR <bench.py args...>
set -e
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/$1; shift
rm -rf "$OUT"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --output-format csv --pmc SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES -d "$OUT/sq1_X" -o p -- python3 $REPO/bench.py "$@" > "$OUT/bench_sq1_X.json" 2> "$OUT/sq1.err"
rocprofv3 --output-format csv --pmc GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_SCA -d "$OUT/sq2_X" -o p -- python3 $REPO/bench.py "$@" > "$OUT/bench_sq2_X.json" 2> "$OUT/sq2.err"

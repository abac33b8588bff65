#!/bin/bash
# Collects the rocprofv3 evidence bench.py's roofline object cites, on the GPU box:
#   gpurun --timeout 1100 -- 'bash tools/collect_profiles.sh gpurun_out/final'
# optional: a list of generators and a list of passes, e.g.  ... gpurun_out/x "Rectilinear" "sq"   (default: all / "bench trace hbm sq mix")
# then, back in the container:  python3 tools/summarize_profiles.py gpurun_out/final profiles/r01 v4
# Every pass runs the bench workload (3 timed frames) of ONE generator, so that a kernel shared between generators (the Fast
# intersect scan also serves the interpolating lattice) is averaged over one use only; counters are collected in their own
# passes, never together with a trace (MI355X_MICROARCH.md, HBM/rocprofv3 section).
set -e
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/${1:-gpurun_out/final}
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
GENS=${2:-Rectilinear Fast InterpolatingRectilinear}
PASSES=${3:-bench trace hbm sq mix}
has() { [[ " $PASSES " == *" $1 "* ]]; }
if has bench; then
    python3 $REPO/bench.py > "$OUT/bench_default.json" 2> "$OUT/bench_default.err"
    echo "[profiles] default bench done"
fi
SQ1="SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_INSTS_SALU SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_THREAD_CYCLES_VALU SQ_WAVES SQ_WAVE_CYCLES"
SQ2="GRBM_GUI_ACTIVE SQ_ACTIVE_INST_ANY SQ_INSTS_SMEM SQ_INSTS_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY"
# the instruction mix behind the roofline's flop figure: FP64 add / mul / fma / transcendental (v_rcp_f64 ...), integer, conversions
MIX1="SQ_INSTS_VALU SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_CVT"
MIX2="SQ_INSTS_VALU_FLOPS_FP64 SQ_INSTS_VALU_FLOPS_FP64_TRANS SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_TRANS_F32 SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_VALU"
for GEN in $GENS; do
    BENCH="$REPO/bench.py --steps 3 --warmup 1 --no-cpu-baseline --only --generator $GEN"
    if has trace; then
        rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_$GEN" -o t -- python3 $BENCH > "$OUT/bench_trace_$GEN.json" 2> "$OUT/trace_$GEN.err"
        echo "[profiles] $GEN kernel trace done"
    fi
    [ $GEN = InterpolatingRectilinear ] && continue
    if has hbm; then
    rocprofv3 --output-format csv --pmc FETCH_SIZE -d "$OUT/fetch_$GEN" -o p -- python3 $BENCH > "$OUT/bench_fetch_$GEN.json" 2> "$OUT/fetch_$GEN.err"
    rocprofv3 --output-format csv --pmc WRITE_SIZE -d "$OUT/write_$GEN" -o p -- python3 $BENCH > "$OUT/bench_write_$GEN.json" 2> "$OUT/write_$GEN.err"
    echo "[profiles] $GEN FETCH_SIZE / WRITE_SIZE done"
    fi
    if has sq; then
    rocprofv3 --output-format csv --pmc $SQ1 -d "$OUT/sq1_$GEN" -o p -- python3 $BENCH > "$OUT/bench_sq1_$GEN.json" 2> "$OUT/sq1_$GEN.err"
    rocprofv3 --output-format csv --pmc $SQ2 -d "$OUT/sq2_$GEN" -o p -- python3 $BENCH > "$OUT/bench_sq2_$GEN.json" 2> "$OUT/sq2_$GEN.err"
    echo "[profiles] $GEN SQ passes done"
    fi
    if has mix; then
    rocprofv3 --output-format csv --pmc $MIX1 -d "$OUT/mix1_$GEN" -o p -- python3 $BENCH > "$OUT/bench_mix1_$GEN.json" 2> "$OUT/mix1_$GEN.err"
    rocprofv3 --output-format csv --pmc $MIX2 -d "$OUT/mix2_$GEN" -o p -- python3 $BENCH > "$OUT/bench_mix2_$GEN.json" 2> "$OUT/mix2_$GEN.err"
    echo "[profiles] $GEN instruction-mix passes done"
    fi
done
find "$OUT" -name "*.csv" -size +20M -delete   # keep what travels back small; the per-dispatch CSVs here are a few hundred KB

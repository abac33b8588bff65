#!/bin/bash
# on the GPU box: time the headline frame with each experimental library under atm-raytracer_amd/csrc/dev/
#   gpurun -- 'bash tools/dev_bench.sh name1 name2 ...'   (extra bench.py flags via DEV_BENCH_FLAGS)
cd "$(dirname "$0")/.."
mkdir -p gpurun_out/dev
for n in "$@"; do
  ATMRT_LIB=$PWD/atm-raytracer_amd/csrc/dev/$n/libatmrt.so python bench.py --steps 3 --warmup 1 --no-cpu-baseline --only $DEV_BENCH_FLAGS > gpurun_out/dev/$n.json 2> gpurun_out/dev/$n.err
  python - "$n" <<'PY'
import json, sys
n = sys.argv[1]
try:
    d = json.load(open(f"gpurun_out/dev/{n}.json"))
    print(f"{n:24s} {d['ms_per_step']:9.3f} ms/frame  {d['value']:.4g} ray-steps/s  kernel {d['roofline']['kernel']} {d['roofline']['kernel_ms']:.3f} ms")
except Exception as e:
    print(n, "FAILED", e)
PY
done

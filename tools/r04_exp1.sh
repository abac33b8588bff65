#!/bin/bash
# round-4 experiment: tight segments on/off on ONE box, Fast after the frozen finished rows, object scenes through the queue tracer
cd ${GRAFT_REPO_ROOT:-/root/repo}
O=gpurun_out/r04_exp1; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --only --steps 6 --warmup 2"
j() { python3 -c "import json,sys;d=json.load(open(sys.argv[1]));print(sys.argv[1].split('/')[-1], round(d['ms_per_step'],2), d['roofline']['phase_ms'])" $1; }
$B > $O/head_tight.json 2> $O/head_tight.err && j $O/head_tight.json
ATMRT_NO_TIGHT=1 $B > $O/head_notight.json 2> $O/head_notight.err && j $O/head_notight.json
$B > $O/head_tight2.json 2> $O/head_tight2.err && j $O/head_tight2.json
$B --generator Fast > $O/fast.json 2> $O/fast.err && j $O/fast.json
timeout -k 10 600 python3 -m pytest tests/test_golden.py tests/test_gpu_march_variants.py tests/test_gpu_parity.py -x -q -k "not random" > $O/parity.log 2>&1; tail -3 $O/parity.log
timeout -k 10 300 $B --objects 1000 --terrain-alpha 0.5 > $O/c5_rect.json 2> $O/c5_rect.err && j $O/c5_rect.json
ATMRT_TRACE_QUEUE_WGS=512 timeout -k 10 300 $B --objects 1000 --terrain-alpha 0.5 > $O/c5_rect_512.json 2> $O/c5_rect_512.err && j $O/c5_rect_512.json
ATMRT_TRACE_QUEUE_WGS=128 timeout -k 10 300 $B --objects 1000 --terrain-alpha 0.5 > $O/c5_rect_128.json 2> $O/c5_rect_128.err && j $O/c5_rect_128.json
timeout -k 10 400 python3 tools/measure_shard_balance.py 8 alpha=0.5 objects=1000 > $O/shard_c5.json 2> $O/shard_c5.err; tail -2 $O/shard_c5.err

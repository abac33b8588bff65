#!/bin/bash
# config 5 (Rectilinear, 1000 objects, terrain_alpha 0.5): kernel trace + stats
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
O=$R/gpurun_out/r04_c5prof; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace -o t -- python3 $R/bench.py --no-cpu-baseline --only --steps 3 --warmup 1 --objects 1000 --terrain-alpha 0.5 > $O/bench.json 2> $O/bench.err
python3 - <<PY
import csv,glob
f=glob.glob("$O/trace/**/t_kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:8]: print(r["Name"][:60], r["Calls"], round(float(r["AverageNs"])/1e6,2), "ms")
PY
cd $R; python3 - <<PY
import sys; sys.path.insert(0,"tests")
from atm_raytracer_amd import generators, synth
from util import run_gpu, frame_stats
ctx=generators.Context(0)
tiles=synth.scene("headline",level=1)[1]
cfg=synth.scene("headline",generator="Rectilinear",terrain_alpha=0.5)[0]; synth.add_objects(cfg)
r=run_gpu(ctx,cfg,tiles); print("stats",frame_stats(ctx),"ray_steps",r["ray_steps"])
PY

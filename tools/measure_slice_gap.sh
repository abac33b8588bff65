#!/bin/bash
# The host round trip between the two kernels of the time-sliced march (k_rect_march_first -> read the number of groups still
# marching -> k_rect_march_cont): how long is the GPU idle?  rocprofv3 kernel trace of tile 3 of 8 of the headline, 6 frames.
#   gpurun -- 'bash tools/measure_slice_gap.sh gpurun_out/slice_gap'
set -e
REPO=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
OUT=$REPO/${1:-gpurun_out/slice_gap}; mkdir -p "$OUT"
cat > /tmp/one_tile.py <<PY
import sys; sys.path.insert(0, "$REPO")
import torch
from atm_raytracer_amd import generators, synth
W, H, G, g = 4096, 2048, 8, 3
cfg, tiles = synth.scene("headline", W, H, generator="Rectilinear", level=2)
ctx = generators.Context(0)
terrain = generators.Terrain.from_tiles(tiles, ctx)
cfg.params.col_begin, cfg.params.col_end = g * W // G, (g + 1) * W // G
_, pod = generators.image_planes(H, W // G, torch.device("cuda", 0))
gen = generators.make_generator(generators.Params(cfg), terrain)
for _ in range(6):
    print(gen.generate_device(pod))
PY
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d "$OUT/trace" -o t -- python3 /tmp/one_tile.py > "$OUT/run.txt" 2> "$OUT/run.err"
python3 - <<PY
import csv, glob, json
f = glob.glob("$OUT/trace/**/t_kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
first = [r for r in rows if "k_rect_march_first" in r["Kernel_Name"]]
cont = [r for r in rows if "k_rect_march_cont" in r["Kernel_Name"]]
fin = [r for r in rows if "k_rect_finalize" in r["Kernel_Name"]]
out = []
for a, b, c in zip(first, cont, fin):
    out.append({"first_ms": (int(a["End_Timestamp"]) - int(a["Start_Timestamp"])) / 1e6,
                "gap_first_to_cont_ms": (int(b["Start_Timestamp"]) - int(a["End_Timestamp"])) / 1e6,
                "cont_ms": (int(b["End_Timestamp"]) - int(b["Start_Timestamp"])) / 1e6,
                "gap_cont_to_finalize_ms": (int(c["Start_Timestamp"]) - int(b["End_Timestamp"])) / 1e6})
json.dump(out, open("$OUT/slice_gap.json", "w"), indent=1)
for o in out[1:]:
    print({k: round(v, 3) for k, v in o.items()})
PY

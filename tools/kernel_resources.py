#!/usr/bin/env python3
"""Print VGPR / SGPR / scratch / occupancy per kernel of atmrt_kernels.hip (hipcc -Rpass-analysis)."""
import re
import subprocess
import sys
import os

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
units = [a for a in sys.argv[1:] if a.endswith(".hip")] or ["atmrt_kernels.hip", "atmrt_paths.hip", "atmrt_march_linear.hip"]
flags = [a for a in sys.argv[1:] if not a.endswith(".hip")]
out = ""
for unit in units:
    src = os.path.join(root, "atm-raytracer_amd", "csrc", unit)
    # the per-unit flags of csrc/Makefile (MARCH_EXTRA, CALL_EXTRA): without them the numbers are not those of the shipped library
    extra = []
    if unit.startswith(("atmrt_march_", "atmrt_trace_")):
        extra += ["-mllvm", "-disable-machine-licm"]
    if unit.startswith("atmrt_trace_") or unit == "atmrt_kernels.hip":
        extra += ["-mllvm", "-enable-ipra=0"]
    cmd = ["/opt/rocm/bin/hipcc", "-std=c++17", "-O3", "-fPIC", "-ffp-contract=off", "-fno-fast-math", "--offload-arch=gfx950", "-c", src,
           "-o", "/tmp/atmrt_k.o", "-Rpass-analysis=kernel-resource-usage"] + extra + flags
    out += subprocess.run(cmd, capture_output=True, text=True).stderr
cur, rows = None, {}
for line in out.splitlines():
    m = re.search(r"Function Name: (\S+)", line)
    if m:
        cur = subprocess.run(["c++filt", m.group(1)], capture_output=True, text=True).stdout.strip().split("(")[0]
        rows[cur] = {}
        continue
    m = re.search(r"remark:\s+([A-Za-z][^:]*): (\S+)", line)
    if m and cur:
        rows[cur][m.group(1).strip()] = m.group(2)
for k, v in rows.items():
    print(f"{k:42s} VGPR {v.get('VGPRs','?'):>4} AGPR {v.get('AGPRs','?'):>3} SGPR {v.get('TotalSGPRs','?'):>4} "
          f"scratch {v.get('ScratchSize [bytes/lane]','?'):>5} occ {v.get('Occupancy [waves/SIMD]','?'):>2} "
          f"LDS {v.get('LDS Size [bytes/block]','?')}")

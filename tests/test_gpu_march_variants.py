"""k_rect_march exists in two builds — the plain one (4 wavefronts per SIMD, no scratch: launches above 16384 workgroups, i.e. the
headline frame) and the small-launch one (wave priorities that fall with progress, 5 wavefronts per SIMD: multi-GPU shards, test
frames) — picked by the size of the launch (atmrt_march_impl.h, ATMRT_LAUNCH_MARCH).  Every test frame below 4 Mpixel runs the second
one, so this test forces each variant in a child process (ATMRT_MARCH_VARIANT is read once per process) and requires the same bits
from both for opaque, translucent and object scenes."""
import hashlib
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import hashlib, json, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
import numpy as np
from atm_raytracer_amd import generators, synth
from util import run_gpu, FIELDS_PIXEL, FIELDS_HIT, bits
ctx = generators.Context(0)
out = {{}}
for name, kw, objects in (("opaque", dict(), False), ("translucent", dict(terrain_alpha=0.5), False), ("objects", dict(terrain_alpha=0.5), True)):
    cfg, tiles = synth.scene("S2", 160, 96, generator="Rectilinear", max_distance=60_000.0, tilt=-1.0, **kw)
    if objects:
        synth.add_objects(cfg, n_cyl=40, n_bill=10, dist=(300.0, 20_000.0), spread_deg=30.0, radius=(30.0, 120.0), height=(150.0, 600.0),
                          bill_w=(150.0, 500.0), bill_h=(150.0, 500.0))
    r = run_gpu(ctx, cfg, tiles)
    h = hashlib.sha256()
    for k in FIELDS_PIXEL + FIELDS_HIT:
        h.update(np.ascontiguousarray(bits(r[k])).tobytes())
    out[name] = [h.hexdigest(), int(r["n_hits"]), int(r["ray_steps"])]
print("RESULT " + json.dumps(out))
"""


def _run(variant):
    env = dict(os.environ)
    if variant:
        env["ATMRT_MARCH_VARIANT"] = variant
    else:
        env.pop("ATMRT_MARCH_VARIANT", None)
    p = subprocess.run([sys.executable, "-c", CHILD.format(root=ROOT, tests=os.path.join(ROOT, "tests"))], env=env, capture_output=True,
                       text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [l for l in p.stdout.splitlines() if l.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.mark.gpu
def test_both_march_variants_produce_the_same_frames():
    plain, small = _run("plain"), _run("small")
    assert plain == small
    assert plain["opaque"][1] > 1000 and plain["translucent"][1] > plain["opaque"][1] and plain["objects"][1] > 0

"""The Rectilinear march exists in three builds — the plain k_rect_march (4 wavefronts per SIMD, no scratch: launches above 16384
workgroups, i.e. the headline frame), the small-launch one (wave priorities that fall with progress, 5 wavefronts per SIMD: shards
and test frames with translucent terrain or objects) and, for small launches over opaque terrain, the time-sliced march
(k_rect_march_first / _cont: ray state through HBM between slices of 128 steps, a FIFO of ray groups) — picked by the size and kind
of the launch (atmrt_kernels.h march_slice_layout, atmrt_march_impl.h ATMRT_LAUNCH_MARCH).  Every test frame below 4 Mpixel runs the
second or third, so this test forces each variant in a child process (ATMRT_MARCH_VARIANT is read once per process) and requires
the same bits from all for opaque, translucent and object scenes; the opaque scene comes in a second size whose pixel count is not
a multiple of a workgroup, and with a step that gives rays of 18 slices."""
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
from util import run_gpu, FIELDS_PIXEL, FIELDS_HIT, bits, frame_stats
ctx = generators.Context(0)
out = {{}}
for name, kw, objects, size in (("opaque", dict(), False, (160, 96)), ("translucent", dict(terrain_alpha=0.5), False, (160, 96)),
                                ("objects", dict(terrain_alpha=0.5), True, (160, 96)), ("opaque-ragged", dict(step=26.0), False, (150, 61))):
    cfg, tiles = synth.scene("S2", size[0], size[1], generator="Rectilinear", max_distance=60_000.0, tilt=-1.0, **kw)
    if objects:
        synth.add_objects(cfg, n_cyl=40, n_bill=10, dist=(300.0, 20_000.0), spread_deg=30.0, radius=(30.0, 120.0), height=(150.0, 600.0),
                          bill_w=(150.0, 500.0), bill_h=(150.0, 500.0))
    r = run_gpu(ctx, cfg, tiles)
    h = hashlib.sha256()
    for k in FIELDS_PIXEL + FIELDS_HIT:
        h.update(np.ascontiguousarray(bits(r[k])).tobytes())
    out[name] = [h.hexdigest(), int(r["n_hits"]), int(r["ray_steps"]), int(frame_stats(ctx)["retraced_pixels"])]
print("RESULT " + json.dumps(out))
"""


def _run(variant, **extra):
    env = dict(os.environ, **extra)
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
def test_all_march_variants_produce_the_same_frames():
    plain, small, sliced, default = _run("plain"), _run("small"), _run("sliced"), _run(None)
    assert plain == small == sliced == default
    assert plain["opaque"][1] > 1000 and plain["translucent"][1] > plain["opaque"][1] and plain["objects"][1] > 0
    assert plain["opaque-ragged"][1] > 500
    # Trace points beyond a pixel's four slots travel through the overflow arena; with an arena of 4 records (or none) they come
    # from a second pass over those pixels instead — the route of rounds 1-2, still the fall-back.  Same bits.
    assert plain["translucent"][3] > 4 and plain["objects"][3] > 4, "the scenes must have pixels beyond the slots"
    assert _run(None, ATMRT_OVERFLOW_CAP="4") == plain and _run("sliced", ATMRT_OVERFLOW_CAP="0") == plain


@pytest.mark.gpu
def test_general_tracer_on_the_probe_scenes_against_the_oracle():
    """The witness scenes of profiles/r04/ipra/README.md through k_rect_trace (ATMRT_MARCH_VARIANT=plain: a small frame otherwise
    takes the small-launch march and never reaches the tracer), every field AND the ray-step count against the oracle.  These are
    the scenes on which a tracer built with the default register allocator is wrong — with IPRA on (656 of 993 hits) and, capped at
    128 VGPRs, with IPRA off (ray-step counts beyond what a ray can have) — while the variant test above still passes."""
    env = dict(os.environ, ATMRT_MARCH_VARIANT="plain")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "trace_waves_probe.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith(("scene_objects", "per_lane_list", "one_sided"))]
    assert len(lines) == 3 and all(l.endswith("identical") for l in lines), p.stdout

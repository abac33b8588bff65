"""The deterministic-math flavour of the oracle against the glibc-libm flavour (what the Rust reference calls):
identical hit/miss and trace-point counts, every field within 1e-9 relative.  This is the evidence that comparing
the GPU bit-for-bit with `det` says something about the reference's libm-based arithmetic."""
import numpy as np
import pytest

from atm_raytracer_amd import synth
from util import run_oracle

CASES = [("S1", 64, 32, "Fast", {}), ("S1", 32, 16, "Rectilinear", {}), ("S2", 96, 48, "Fast", {}), ("S2", 32, 20, "Rectilinear", {}),
         ("S2", 48, 24, "Fast", {"earth_shape": "Wgs84"}), ("S2", 48, 24, "Fast", {"terrain_alpha": 0.5, "tilt": -4.0}),
         ("S2", 40, 20, "InterpolatingRectilinear", {"max_distance": 80_000.0}), ("S3", 64, 32, "Fast", {})]


def test_det_matches_libm_with_frusta(oracle_det, oracle_libm):
    """Scene objects with constant colours (cylinders / cones / frusta): same trace points in both flavours.  (Billboards are
    excluded on purpose: their texel blends are truncated to u8, where a last-bit difference can change a level.)"""
    cfg, tiles = synth.scene("S2", 48, 24, generator="Fast", terrain_alpha=0.5, tilt=-2.0, max_distance=30_000.0)
    synth.add_objects(cfg, n_cyl=36, n_bill=0, dist=(300.0, 6_000.0), spread_deg=28.0, radius=(30.0, 120.0), height=(150.0, 600.0))
    a, b = run_oracle(oracle_det, cfg, tiles), run_oracle(oracle_libm, cfg, tiles)
    assert (a["color_tag"] == 1).sum() > 50
    assert np.array_equal(a["hit_count"], b["hit_count"]) and np.array_equal(a["color_tag"], b["color_tag"])
    for k in ("lat", "lon", "distance", "elevation", "path_length", "normal", "rgba"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-9, atol=1e-9, err_msg=k)


@pytest.mark.parametrize("scene,w,h,gen,kw", CASES)
def test_det_matches_libm(oracle_det, oracle_libm, scene, w, h, gen, kw):
    cfg, tiles = synth.scene(scene, w, h, generator=gen, **kw)
    a, b = run_oracle(oracle_det, cfg, tiles), run_oracle(oracle_libm, cfg, tiles)
    assert np.array_equal(a["hit_count"], b["hit_count"]) and a["ray_steps"] == b["ray_steps"]
    for k in ("azimuth", "elevation_angle", "lat", "lon", "distance", "elevation", "path_length"):
        np.testing.assert_allclose(a[k], b[k], rtol=1e-9, atol=1e-9, err_msg=k)
    np.testing.assert_allclose(a["normal"], b["normal"], rtol=1e-9, atol=1e-9)

"""The product's numerics header (csrc/atmrt_core.h, the device functions of the HIP kernels) compiled for the
host — test-only — and compared BIT-FOR-BIT with the oracle's `det` flavour.  This pins the operation order of
the two independent restatements on the CPU, so a GPU run only has to confirm that gfx950 executes the same
IEEE operations (tests/test_gpu_parity.py)."""
import ctypes as C

import numpy as np
import pytest

import cbuild
from atm_raytracer_amd import _abi, config, synth

EARTHS = ["SimpleSphere", {"Spherical": {"radius": 6371000.0}}, {"Ellipsoid": {"a": 6378137.0, "b": 6356752.3}}, "Wgs84",
          "AzimuthalEquidistant", "FlatDistorted", {"ObserverAe": {"proj_radius": 6400000.0}}, "SimpleObserverAe"]


@pytest.fixture(scope="module")
def core():
    return C.CDLL(cbuild.core_host())


def ptr(a):
    return C.c_void_p(a.ctypes.data)


def test_atmosphere_and_refractive_index(core, oracle_det):
    atm = oracle_det.us76()
    env = oracle_det.env(atm, 530e-9)
    h = np.concatenate([np.linspace(-900, 80000, 4001), [10999.99, 11000.0, 11000.01, 20000.0, 47000.0]])
    t, p, n, dn = (np.empty_like(h) for _ in range(4))
    assert core.ch_atm(C.byref(atm), C.c_double(530e-9), C.c_size_t(h.size), ptr(h), ptr(t), ptr(p), ptr(n), ptr(dn)) == 0
    for i, hi in enumerate(h):
        assert t[i] == oracle_det.temperature(env, hi) and p[i] == oracle_det.pressure(env, hi)
        assert n[i] == oracle_det.n(env, hi) and dn[i] == oracle_det.dn(env, hi)


@pytest.mark.parametrize("earth", EARTHS)
def test_geodesy(core, oracle_det, earth):
    e = config._earth(earth)
    rng = np.random.default_rng(2)
    for lat0, lon0, dr in [(46.5, 8.5, 0.0), (46.5, 8.5, 90.0), (-33.2, 151.1, 237.3), (0.5, -0.5, 359.0), (70.0, 179.9, 45.0)]:
        d = np.concatenate([[0.0, 15.0, -15.0], rng.uniform(0, 400e3, 200)])
        lat, lon = np.empty_like(d), np.empty_like(d)
        assert core.ch_coords(C.byref(e), C.c_double(lat0), C.c_double(lon0), C.c_double(dr), C.c_size_t(d.size), ptr(d), ptr(lat), ptr(lon)) == 0
        want = oracle_det.coords_at_dist(e, lat0, lon0, dr, d)
        assert np.array_equal(lat, want[:, 0]) and np.array_equal(lon, want[:, 1])
        out = np.empty(12)
        assert core.ch_cart(C.byref(e), C.c_double(lat0), C.c_double(lon0), C.c_double(123.4), ptr(out)) == 0
        n, ea, up = oracle_det.world_directions(e, lat0, lon0)
        assert np.array_equal(out, np.concatenate([oracle_det.as_cartesian(e, lat0, lon0, 123.4), n, ea, up]))


@pytest.mark.parametrize("earth", ["SimpleSphere", "FlatDistorted", "Wgs84"])
@pytest.mark.parametrize("straight", [0, 1])
def test_ray_stepper(core, oracle_det, earth, straight):
    cfg, _ = synth.scene("S2", 8, 8, earth_shape=earth)
    for ang in (-2.0, -0.3, 0.0, 0.7, 12.0):
        x, h = np.empty(801), np.empty(801)
        assert core.ch_ray_path(C.byref(cfg.atmosphere), C.byref(cfg.params.earth), C.c_double(530e-9), C.c_double(1500.0), C.c_double(ang),
                                C.c_int(straight), C.c_double(75.0), C.c_size_t(800), ptr(x), ptr(h)) == 0
        xo, ho = oracle_det.ray_paths(cfg.params, 1500.0, [ang], 75.0, 800, bool(straight), cfg.atmosphere)
        assert np.array_equal(x, xo[0]) and np.array_equal(h, ho[0])


@pytest.mark.parametrize("earth", ["SimpleSphere", "Wgs84", "AzimuthalEquidistant", "FlatDistorted"])
def test_terrain_sampler_and_normals(core, oracle_det, earth):
    e = config._earth(earth)
    posts = synth.synth_tile(46, 8, level=1)
    t = oracle_det.terrain_new({(46, 8): posts})
    rng = np.random.default_rng(4)
    lat = np.concatenate([rng.uniform(45.95, 47.05, 400), [46.0, 47.0, 46.5, 46.99999999999999]])
    lon = np.concatenate([rng.uniform(7.95, 9.05, 400), [8.0, 9.0, 8.25, 8.99999999999999]])
    elev, valid, nrm = np.empty_like(lat), np.empty(lat.size, dtype=np.int32), np.empty((lat.size, 3))
    assert core.ch_terrain(C.byref(e), 46, 8, posts.shape[0], posts.shape[1], ptr(posts), C.c_size_t(lat.size), ptr(lat), ptr(lon), ptr(elev),
                           ptr(valid), ptr(nrm)) == 0
    for i in range(lat.size):
        want = oracle_det.get_elev(t, float(lat[i]), float(lon[i]))
        assert bool(valid[i]) == (want is not None)
        if want is not None:
            assert elev[i] == want
        assert np.array_equal(nrm[i], oracle_det.find_normal(e, t, float(lat[i]), float(lon[i])))
    oracle_det.terrain_free(t)


def test_pixel_to_ray_mapping(core, oracle_det):
    for gen, tilt, direction in (("Fast", -3.0, 350.0), ("Rectilinear", 7.5, 123.0)):
        cfg, _ = synth.scene("S1", 37, 22, generator=gen, tilt=tilt, direction=direction, fov=77.0, max_distance=300.0)
        p = cfg.params
        fd, fe = np.empty(p.width), np.empty(p.height)
        rd, re = np.empty((p.height, p.width)), np.empty((p.height, p.width))
        assert core.ch_pixels(C.byref(p), ptr(fd), ptr(fe), ptr(rd), ptr(re)) == 0
        res = oracle_det.generate(p)
        if gen == "Fast":
            az = np.where(fd < 0, fd + 360.0, np.where(fd >= 360.0, fd - 360.0, fd))
            assert np.array_equal(res["azimuth"], np.broadcast_to(az, (p.height, p.width)))
            assert np.array_equal(res["elevation_angle"], np.broadcast_to(fe[:, None], (p.height, p.width)))
        else:
            assert np.array_equal(res["azimuth"], rd * (180.0 / np.pi)) and np.array_equal(res["elevation_angle"], re * (180.0 / np.pi))

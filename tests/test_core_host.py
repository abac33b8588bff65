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


WILD_SPLINE = {"pressure": {"altitude": 0.0, "pressure": 102390.63927278577},
               "first_temperature_function": {"Spline": {"boundary_condition": "Natural", "points": [
                   [-500.0, 295.22010975963303], [16672.2152178729, 192.93808594359285], [16688.49864235047, 195.36762037322703],
                   [21728.827855811145, 170.01112581350702], [28892.825051123004, 125.19791348512327]]}}}


def _certify(core, atm, spherical=True, radius=6371000.0, step=50.0, wavelength=530e-9):
    n = C.c_int(0)
    cap = sum(atm.functions[j].n_points + 1 if atm.functions[j].kind == 1 else 1 for j in range(atm.n_functions))  # most segments the table can have
    arrs = [np.zeros(cap) for _ in range(15)]
    band = np.zeros(2)
    assert core.ch_certify(C.byref(atm), C.c_double(wavelength), int(spherical), C.c_double(radius), C.c_double(step), C.byref(n),
                           ptr(arrs[0]), ptr(arrs[1]), ptr(arrs[2]), ptr(band), ptr(arrs[3]), ptr(arrs[4]), ptr(arrs[5]), ptr(arrs[6]), ptr(arrs[7]),
                           ptr(arrs[8]), ptr(arrs[9]), ptr(arrs[10]), ptr(arrs[11]), ptr(arrs[12]), ptr(arrs[13]), ptr(arrs[14])) == 0
    keys = ("from", "safe_lo", "safe_hi", "min_t", "max_pt", "max_z_dev", "max_n", "max_abs_e", "flags", "max_dt_rel", "max_dz_rel", "tight_lo",
            "tight_hi", "tight_max_q", "tight_max_zdev")
    return {k: a[:n.value] for k, a in zip(keys, arrs)}, band


def test_certified_intervals_cover_the_standard_atmosphere(core):
    """atm_certify (atmrt_core.h): the GPU takes its shortcut divisions (dm_div, dm_div3, dm_sqrt_inrange: IEEE results for in-range
    operands only) inside [safe_lo, safe_hi) of a segment and IEEE operations outside.  US-76: every layer is certified from its lower
    to its upper boundary — the marching kernels never leave the fast path below the point where the last layer's extrapolated
    temperature reaches 1 K (178 km) — and the lowest layer down to where Z leaves [0.5, 1.5] (-47 km)."""
    c, band = _certify(core, config.us76())
    bounds = [11000.0, 20000.0, 32000.0, 47000.0, 51000.0, 71000.0]
    assert list(c["safe_hi"][:6]) == bounds and list(c["safe_lo"][1:]) == bounds
    assert -60000.0 < c["safe_lo"][0] < -30000.0 and 170000.0 < c["safe_hi"][6] < 180000.0
    # Round 4: every layer of the standard atmosphere has a TIGHT part (flag 2; flag 1 marks the isothermal ones) on which the
    # kernels' three-point divisions run without votes and with seeded reciprocals: the whole certified interval of the middle
    # layers; the lowest layer (certified down to -47 km) from -1500 m up — no ray marches below -1000 m —, the highest (certified up
    # to 178 km, where its extrapolated temperature reaches 1 K) while T is above ~40 K, to ~157 km.
    assert [int(f) for f in c["flags"]] == [2, 3, 2, 2, 3, 2, 2]
    assert list(c["tight_lo"][1:]) == bounds and list(c["tight_hi"][:6]) == bounds
    assert c["tight_lo"][0] == -1500.0 and 150000.0 < c["tight_hi"][6] < 160000.0
    assert c["max_dt_rel"].max() <= 2.0 ** -21 and c["max_dz_rel"].max() <= 2.0 ** -21
    assert c["tight_max_q"].max() <= 2.0 ** -10.5 and c["tight_max_zdev"].max() <= 2.0 ** -10.5
    # a polar-winter surface layer (-40 C at 1040 hPa): certified as before, not tight (|1 - Z| = 1.3e-3) — the voting path serves it
    cold = config.Config.from_dict({"atmosphere": {"pressure": {"altitude": 0.0, "pressure": 104000.0},
                                                   "temperature_fixed_point": {"altitude": 0.0, "temperature": 233.15},
                                                   "first_temperature_function": {"Linear": {"gradient": 0.002}}},
                                    "output": {"width": 8, "height": 8}})
    cc, _ = _certify(core, cold.atmosphere)
    assert int(cc["flags"][0]) == 0 and cc["tight_lo"][0] == np.inf and cc["safe_lo"][0] < 1.0 < 1000.0 < cc["safe_hi"][0] and cc["max_z_dev"][0] > 2.0 ** -10.5
    assert band[0] == -100000.0 and band[1] == 1.0e7
    # an independent dense sample of the certified intervals stays far inside the shortcuts' range
    assert c["min_t"].min() >= 1.0 and c["max_z_dev"].max() <= 0.5 and c["max_n"].max() < 2.0 and c["max_pt"].max() < 2.0e5
    assert c["max_abs_e"].max() <= 650.0  # exp's main branch (|e| <= 700) needs no range vote inside a certified interval
    # a planet of 10 km: the band ends 1 km above its centre; a radius, step or wavelength outside the supported span: nothing certified
    c, band = _certify(core, config.us76(), radius=10000.0)
    assert band[0] == -9000.0 and c["safe_lo"][0] == -9000.0
    for kw in ({"radius": 10.0}, {"step": 1e-6}, {"step": 1e12}, {"radius": 1e300}):
        c, band = _certify(core, config.us76(), **kw)
        assert np.all(c["safe_lo"] == np.inf) and np.all(c["safe_hi"] == -np.inf) and band[0] == np.inf
    c, band = _certify(core, config.us76(), spherical=False, radius=0.0)  # flat earth shapes have no radius to respect
    assert list(c["safe_hi"][:6]) == bounds


def test_a_spline_that_overshoots_is_certified_only_where_it_is_tame(core):
    """Two knots 16 m apart with 2.4 K between them make a Natural spline swing below 0 K in the intervals next to them (found by the
    random sweep, seed 4899): T = 36 K and p = 1.7e308 Pa at 15.5 km.  Those knot intervals must carry no certificate (the kernels
    then divide in IEEE there); whatever is certified must pass the independent dense sample."""
    cfg = config.Config.from_dict({"atmosphere": WILD_SPLINE, "output": {"width": 8, "height": 8}})
    c, band = _certify(core, cfg.atmosphere, spherical=False, radius=0.0)
    # segments: continuation below -500 m, four knot intervals, continuation above 28.9 km
    assert len(c["from"]) == 6
    certified = c["safe_lo"] < c["safe_hi"]
    assert c["safe_lo"][1] == -500.0 and 2000.0 < c["safe_hi"][1] < 4000.0, "the first knot interval: only below the point where T falls to 1 K"
    assert not certified[2] and not certified[3] and not certified[4], "their base pressure has overflowed already"
    for k in np.flatnonzero(certified):
        assert c["min_t"][k] >= 1.0 and c["max_z_dev"][k] <= 0.5 and c["max_n"][k] < 2.0 and c["max_abs_e"][k] <= 650.0
    # a physical spline (troposphere + inversion) is certified knot to knot
    tame = {"pressure": {"altitude": 0.0, "pressure": 101325.0}, "first_temperature_function": {"Spline": {"boundary_condition": "Natural",
            "points": [[0.0, 288.0], [1000.0, 283.0], [1200.0, 285.0], [5000.0, 260.0], [11000.0, 217.0], [20000.0, 217.0]]}}}
    cfg = config.Config.from_dict({"atmosphere": tame, "output": {"width": 8, "height": 8}})
    c, _ = _certify(core, cfg.atmosphere)
    assert list(c["safe_lo"][1:6]) == [0.0, 1000.0, 1200.0, 5000.0, 11000.0] and list(c["safe_hi"][1:6]) == [1000.0, 1200.0, 5000.0, 11000.0, 20000.0]


@pytest.mark.parametrize("block", range(6))
def test_certificates_of_random_atmospheres_hold_on_a_dense_sample(core, block):
    """atm_certify's bounds are derived (end points of monotone quantities, stationary points of the cubic); ch_certify re-evaluates
    T, p/T, Z and n at 4001 points of every certified interval.  50 random atmospheres per block — splines that overshoot, lapse
    rates up to +-50 K/km, pressures from 200 to 1100 hPa: whatever carries a certificate must be far inside the shortcuts' range
    (T >= 1 K, |Z - 1| <= 1/2, 1 <= n <= 2^8 + 1), and an interval never leaves its segment or the global altitude band."""
    rng = np.random.default_rng(4242 + block)
    n_certified = 0
    n_tight = [0]
    for _ in range(50):
        if rng.uniform() < 0.5:
            n_knots = int(rng.integers(2, 9))
            knots = np.sort(rng.uniform(-1000.0, 40_000.0, n_knots))
            if np.min(np.diff(knots)) <= 0.0:
                continue
            temps = 288.0 - 0.006 * knots + rng.uniform(-15.0, 15.0, n_knots)
            atm = {"pressure": {"altitude": float(rng.uniform(-200.0, 3000.0)), "pressure": float(rng.uniform(300.0, 1100.0)) * 100.0},
                   "first_temperature_function": {"Spline": {"boundary_condition": "Natural", "points": [[float(a), float(t)] for a, t in zip(knots, temps)]}}}
        else:
            grads = [float(rng.choice([-0.0065, 0.0, 0.05, -0.05, -0.0342, float(rng.uniform(-0.02, 0.02)), 1e-9])) for _ in range(int(rng.integers(1, 7)))]
            alts = np.sort(rng.uniform(-500.0, 50_000.0, len(grads) - 1))
            atm = {"pressure": {"altitude": float(rng.uniform(-300.0, 5000.0)), "pressure": float(rng.uniform(200.0, 1100.0)) * 100.0},
                   "temperature_fixed_point": {"altitude": float(rng.uniform(-300.0, 12000.0)), "temperature": float(rng.uniform(180.0, 330.0))},
                   "first_temperature_function": {"Linear": {"gradient": grads[0]}},
                   "next_functions": [{"altitude": float(a), "function": {"Linear": {"gradient": g}}} for a, g in zip(alts, grads[1:])]}
        cfg = config.Config.from_dict({"atmosphere": atm, "output": {"width": 8, "height": 8}})
        spherical = bool(rng.uniform() < 0.7)
        radius = float(rng.choice([6371000.0, 3.0e6, 2.0e5])) if spherical else 0.0
        c, band = _certify(core, cfg.atmosphere, spherical=spherical, radius=radius, wavelength=float(rng.uniform(300e-9, 1100e-9)))
        n = len(c["from"])
        for k in range(n):
            if not c["safe_lo"][k] < c["safe_hi"][k]:
                continue
            n_certified += 1
            # (the interval ends are found by bisection on the bounds: allow their rounding)
            assert c["min_t"][k] >= 1.0 - 1e-9 and c["max_z_dev"][k] <= 0.5 + 1e-9 and 1.0 <= c["max_n"][k] <= 257.0, (atm, k)
            assert c["max_abs_e"][k] <= 650.0, (atm, k)
            if int(c["flags"][k]) & 2:  # a tight segment keeps its promises on the dense sample too
                n_tight[0] += 1
                assert c["max_dt_rel"][k] <= 2.0 ** -21 * 1.001 and c["max_dz_rel"][k] <= 2.0 ** -21 * 1.001, (atm, k)
                assert c["tight_max_q"][k] <= 2.0 ** -10.5 and c["tight_max_zdev"][k] <= 2.0 ** -10.5, (atm, k)
                assert c["safe_lo"][k] <= c["tight_lo"][k] < c["tight_hi"][k] <= c["safe_hi"][k], (atm, k)
            assert band[0] <= c["safe_lo"][k] and c["safe_hi"][k] <= band[1]
            if k > 0:
                assert c["safe_lo"][k] >= c["from"][k]
            if k + 1 < n:
                assert c["safe_hi"][k] <= c["from"][k + 1]
    assert n_certified > 50 and n_tight[0] > 5, (n_certified, n_tight)

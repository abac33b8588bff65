"""`Spline` temperature functions (schema: reference README.md:283-323).  The crate that implements them
in the reference is absent, so the restatement (interpolating cubic spline + Gauss-Legendre hydrostatic quadrature) is pinned
to closed forms here; the product header and the GPU are then compared with the oracle bit for bit."""
import ctypes as C
import math

import numpy as np
import pytest
import yaml

import cbuild
from atm_raytracer_amd import _abi, config, generators, synth
from util import assert_bitexact, run_gpu, run_oracle

INVERSION = """
atmosphere:
    pressure: {altitude: 0.0, pressure: 101325}
    first_temperature_function:
        Spline:
            boundary_condition: Natural
            points: [[0.0, 283.15], [40.0, 283.6], [80.0, 287.9], [150.0, 289.2], [400.0, 287.4], [1500.0, 280.2]]
    next_functions:
        - {altitude: 1500.0, function: {Linear: {gradient: -0.0065}}}
        - {altitude: 11000.0, function: {Linear: {gradient: 0.0}}}
"""


def atm_from(doc):
    return config.Config.from_dict(yaml.safe_load(doc)).atmosphere


def test_spline_through_collinear_points_is_the_linear_atmosphere(oracle_det):
    """A clamped spline through points on the US-76 troposphere line must reproduce it: same T, same p (quadrature vs
    closed form agree to 1e-13), same n."""
    pts = [[h, 288.15 - 0.0065 * h] for h in (0.0, 500.0, 2000.0, 5000.0, 11000.0)]
    doc = {"atmosphere": {"pressure": {"altitude": 0.0, "pressure": 101325.0},
                          "first_temperature_function": {"Spline": {"boundary_condition": {"Derivatives": [-0.0065, -0.0065]}, "points": pts}},
                          "next_functions": [{"altitude": 11000.0, "function": {"Linear": {"gradient": 0.0}}}]}}
    env_s = oracle_det.env(config.Config.from_dict(doc).atmosphere)
    env_l = oracle_det.env()
    assert env_s.n == 6  # continuation below the first knot + 4 knot intervals + the Linear function
    for h in (-200.0, 0.0, 123.4, 499.9, 500.0, 3210.0, 10999.0, 11000.0, 15000.0):
        assert oracle_det.temperature(env_s, h) == pytest.approx(oracle_det.temperature(env_l, h), abs=1e-10)
        assert oracle_det.pressure(env_s, h) == pytest.approx(oracle_det.pressure(env_l, h), rel=1e-12)
        assert oracle_det.n(env_s, h) == pytest.approx(oracle_det.n(env_l, h), rel=1e-14)


def test_spline_interpolates_and_honours_boundary_conditions(oracle_det):
    a = atm_from(INVERSION)
    env = oracle_det.env(a)
    f = a.functions[0]
    for i in range(f.n_points):
        assert oracle_det.temperature(env, f.point_altitude[i]) == pytest.approx(f.point_temperature[i], abs=1e-10)
    d = 1e-3
    curv = lambda h: (oracle_det.temperature(env, h + d) - 2 * oracle_det.temperature(env, h) + oracle_det.temperature(env, h - d)) / d**2
    assert abs(curv(0.0 + 2 * d)) < 1e-4  # Natural: zero second derivative at the first knot
    for h in (40.0, 80.0, 150.0, 400.0):  # C1 and C2 across interior knots
        l = (oracle_det.temperature(env, h) - oracle_det.temperature(env, h - d)) / d
        r = (oracle_det.temperature(env, h + d) - oracle_det.temperature(env, h)) / d
        assert l == pytest.approx(r, abs=1e-4)
    # the Linear function above 1500 m continues from the spline's value there
    assert oracle_det.temperature(env, 1500.0) == pytest.approx(280.2, abs=1e-10)
    assert oracle_det.temperature(env, 2500.0) == pytest.approx(280.2 - 6.5, abs=1e-9)
    # clamped ends
    b = atm_from("atmosphere: {pressure: {altitude: 0, pressure: 101325}, first_temperature_function: {Spline: {boundary_condition: "
                 "{Derivatives: [-0.01, 0.02]}, points: [[0, 288], [100, 287.5], [300, 289]]}}}")
    envb = oracle_det.env(b)
    assert (oracle_det.temperature(envb, d) - oracle_det.temperature(envb, 0.0)) / d == pytest.approx(-0.01, abs=1e-5)
    assert (oracle_det.temperature(envb, 300.0) - oracle_det.temperature(envb, 300.0 - d)) / d == pytest.approx(0.02, abs=1e-5)
    assert (oracle_det.temperature(envb, 400.0) - oracle_det.temperature(envb, 300.0)) / 100.0 == pytest.approx(0.02, abs=1e-12)  # linear continuation


def test_pressure_is_hydrostatic_everywhere(oracle_det):
    """dp/dh = -g0 M p / (R* T) through spline intervals, across knots and across function boundaries; p continuous."""
    env = oracle_det.env(atm_from(INVERSION))
    gmr = 9.80665 * 0.0289644 / 8.31432
    for h in (1.0, 39.9, 40.1, 79.0, 120.0, 399.0, 401.0, 1499.0, 1501.0, 5000.0, 12000.0):
        d = 0.05
        dpdh = (oracle_det.pressure(env, h + d) - oracle_det.pressure(env, h - d)) / (2 * d)
        assert dpdh == pytest.approx(-gmr * oracle_det.pressure(env, h) / oracle_det.temperature(env, h), rel=2e-8)
    for h in (40.0, 80.0, 150.0, 400.0, 1500.0, 11000.0):
        assert oracle_det.pressure(env, h - 1e-9) == pytest.approx(oracle_det.pressure(env, h), rel=1e-12)
    assert oracle_det.pressure(env, 0.0) == pytest.approx(101325.0, rel=1e-15)


def test_inversion_bends_rays_down(oracle_det):
    """A strong low-level inversion (dT/dh > 0) raises -R dn/dh above the standard 0.17, so a horizontal ray launched inside it
    ends lower than in US-76 (looming), and super-refraction k > 1 traps it."""
    inv, std = atm_from(INVERSION), config.us76()
    k_inv = -6371000.0 * oracle_det.dn(oracle_det.env(inv), 60.0)
    assert k_inv > 0.6
    cfg, _ = synth.scene("S2", 8, 8)
    _, h_inv = oracle_det.ray_paths(cfg.params, 60.0, [0.0], 50.0, 400, False, inv)
    _, h_std = oracle_det.ray_paths(cfg.params, 60.0, [0.0], 50.0, 400, False, std)
    assert h_inv[0, -1] < h_std[0, -1] - 5.0


def test_product_core_matches_oracle_bitexact_with_splines(oracle_det):
    core = C.CDLL(cbuild.core_host())
    a = atm_from(INVERSION)
    env = oracle_det.env(a, 530e-9)
    h = np.concatenate([np.linspace(-50, 2000, 1500), [40.0, 80.0, 150.0, 400.0, 1500.0, 39.99, 1499.99, 11000.0, 30000.0]])
    t, p, n, dn = (np.empty_like(h) for _ in range(4))
    ptr = lambda x: C.c_void_p(x.ctypes.data)
    assert core.ch_atm(C.byref(a), C.c_double(530e-9), C.c_size_t(h.size), ptr(h), ptr(t), ptr(p), ptr(n), ptr(dn)) == 0
    for i, hi in enumerate(h):
        assert t[i] == oracle_det.temperature(env, hi) and p[i] == oracle_det.pressure(env, hi)
        assert n[i] == oracle_det.n(env, hi) and dn[i] == oracle_det.dn(env, hi)


@pytest.mark.gpu
@pytest.mark.parametrize("generator,w,h", [("Fast", 96, 64), ("Rectilinear", 40, 32)])
def test_gpu_parity_with_spline_atmosphere(gpu_ctx, oracle_det, generator, w, h):
    cfg, tiles = synth.scene("S2", w, h, generator=generator, tilt=-0.2, fov=8.0, max_distance=80_000.0)
    cfg.atmosphere = atm_from(INVERSION)
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert got["n_hits"] > 0
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_atmosphere(gpu_ctx.handle, C.byref(config.us76())))


# ---- unbounded definitions: `next_functions` and a Spline's `points` are Vecs in the reference (README.md:283-323, params.rs:453-454).
# Until ABI 4 the C ABI held at most 8 functions / 32 knots / 64 compiled segments; a radiosonde-derived spline has more knots.
def radiosonde_spline(n_knots=200, top=30_000.0, seed=11):
    """A sounding-like profile: a knot every ~150 m up to 30 km, the standard lapse structure + inversions and noise."""
    rng = np.random.default_rng(seed)
    alt = np.sort(np.concatenate([[0.0, top], rng.uniform(20.0, top - 20.0, n_knots - 2)]))
    alt += np.arange(n_knots) * 1e-3  # strictly increasing
    std = np.where(alt < 11000.0, 288.15 - 0.0065 * alt, np.where(alt < 20000.0, 216.65, 216.65 + 0.001 * (alt - 20000.0)))
    temp = std + 1.5 * np.sin(alt / 700.0) + rng.normal(0.0, 0.15, n_knots) + 4.0 * np.exp(-((alt - 900.0) / 250.0) ** 2)
    return {"pressure": {"altitude": 120.0, "pressure": 100_150.0},
            "first_temperature_function": {"Spline": {"boundary_condition": "Natural", "points": [[float(a), float(t)] for a, t in zip(alt, temp)]}}}


def forty_linear_layers():
    """40 Linear layers 400 m thick with alternating lapse rates, continued by an isothermal layer: 41 temperature functions."""
    grads = [(-0.0095 if k % 2 == 0 else -0.0035) if k < 28 else (0.0 if k % 3 else 0.0015) for k in range(40)]
    return {"pressure": {"altitude": 0.0, "pressure": 101_000.0}, "temperature_fixed_point": {"altitude": 0.0, "temperature": 290.0},
            "first_temperature_function": {"Linear": {"gradient": grads[0]}},
            "next_functions": [{"altitude": 400.0 * k, "function": {"Linear": {"gradient": g}}} for k, g in enumerate(grads) if k > 0] +
                              [{"altitude": 16_000.0, "function": {"Linear": {"gradient": 0.0}}}]}


def splines_with_knots_outside_their_ranges():
    """20 functions, every third a Spline whose knots do not fit its altitude range: all BELOW the function's start (seed 300601 of the
    round-4 sweep: the continuation above the last knot used to begin at that knot, below the previous segment's start — a table that
    is not ascending, which the oracle's scan from the top and the product's bisection read differently), all ABOVE its end, or
    sticking out on both sides."""
    fns, alts = [], [600.0 * k for k in range(20)]
    for k in range(20):
        lo, t0 = alts[k], 288.0 - 0.006 * alts[k]
        if k % 3 == 1:
            knots = {0: [lo - 900.0, lo - 650.0, lo - 610.0], 1: [lo + 700.0, lo + 900.0], 2: [lo - 300.0, lo + 200.0, lo + 450.0, lo + 1200.0]}[k // 3 % 3]
            fns.append({"Spline": {"boundary_condition": "Natural", "points": [[a, t0 - 0.005 * (a - lo) + 0.4 * ((i * 7) % 3 - 1)] for i, a in enumerate(knots)]}})
        else:
            fns.append({"Linear": {"gradient": [-0.0065, 0.002, -0.0098][k % 3]}})
    return {"pressure": {"altitude": 100.0, "pressure": 100_200.0}, "temperature_fixed_point": {"altitude": 0.0, "temperature": 288.0},
            "first_temperature_function": fns[0], "next_functions": [{"altitude": a, "function": f} for a, f in zip(alts[1:], fns[1:])]}


BIG_ATMOSPHERES = {"spline-200-knots": radiosonde_spline, "linear-41-functions": forty_linear_layers,
                   "splines-off-their-ranges": splines_with_knots_outside_their_ranges}


@pytest.mark.parametrize("name", sorted(BIG_ATMOSPHERES))
def test_big_atmospheres_compile_identically_in_product_and_oracle(oracle_det, name):
    """The product's host code (atm_compile with its bisection atm_layer) against the oracle (linear search) on atmospheres far
    beyond the old capacities: T, p, n and dn/dh identical to the last bit at 6000 altitudes incl. every boundary."""
    core = C.CDLL(cbuild.core_host())
    a = config._atmosphere(BIG_ATMOSPHERES[name]())
    assert (a.n_functions == 1 and a.functions[0].n_points == 200) or a.n_functions in (20, 41)
    env = oracle_det.env(a, 530e-9)
    assert env.n >= 20
    edges = np.array([env.from_[k] for k in range(env.n)])
    assert np.all(np.diff(edges[1:]) >= 0.0), "segment k >= 1 applies from from[k]: the table must ascend"
    h = np.concatenate([np.linspace(-400.0, 33_000.0, 5000), edges, np.nextafter(edges, -np.inf), edges + 0.004, [np.nan, -1e9, 1e9]])
    t, p, n, dn = (np.empty_like(h) for _ in range(4))
    ptr = lambda x: C.c_void_p(x.ctypes.data)
    assert core.ch_atm(C.byref(a), C.c_double(530e-9), C.c_size_t(h.size), ptr(h), ptr(t), ptr(p), ptr(n), ptr(dn)) == 0
    same = lambda x, y: x == y or (x != x and y != y)
    for i, hi in enumerate(h):
        assert same(t[i], oracle_det.temperature(env, hi)) and same(p[i], oracle_det.pressure(env, hi)), hi
        assert same(n[i], oracle_det.n(env, hi)) and same(dn[i], oracle_det.dn(env, hi)), hi


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(BIG_ATMOSPHERES))
@pytest.mark.parametrize("generator,w,h", [("Fast", 96, 48), ("Rectilinear", 40, 24), ("InterpolatingRectilinear", 72, 40)])
def test_gpu_parity_with_big_atmospheres(gpu_ctx, oracle_det, name, generator, w, h):
    """A 200-knot Spline and 41 Linear functions through all three generators on the GPU, bit-exact against the oracle; the
    sampler harness too (every segment is visited)."""
    atm = config._atmosphere(BIG_ATMOSPHERES[name]())
    cfg, tiles = synth.scene("S2", w, h, generator=generator, tilt=-0.5, fov=30.0, max_distance=90_000.0)
    cfg.atmosphere = atm
    got = run_gpu(gpu_ctx, cfg, tiles)
    assert got["n_hits"] > 0
    assert_bitexact(got, run_oracle(oracle_det, cfg, tiles))
    from atm_raytracer_amd import generators
    env = oracle_det.env(atm, cfg.params.wavelength)
    edges = np.array([env.from_[k] for k in range(env.n)])
    alt = np.concatenate([np.linspace(-300.0, 32_000.0, 3001), edges, edges + 0.01, np.nextafter(edges, -np.inf)])
    s = generators.atmosphere_sample(gpu_ctx, alt)
    for i in list(range(0, 3001, 7)) + list(range(3001, alt.size)):
        assert s["temperature"][i] == oracle_det.temperature(env, alt[i]) and s["n"][i] == oracle_det.n(env, alt[i])
        assert s["pressure"][i] == oracle_det.pressure(env, alt[i]) and s["dn_dh"][i] == oracle_det.dn(env, alt[i])
    gpu_ctx.check(gpu_ctx.lib.atmrt_set_atmosphere(gpu_ctx.handle, C.byref(config.us76())))

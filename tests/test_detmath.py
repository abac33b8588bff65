"""detmath.h (the deterministic elementary functions both the oracle's `det` flavour and the HIP kernels use)
against glibc libm and mpmath: <= 1 ulp for sin/cos/atan/asin/exp/log/atan2, <= 2 ulp for tan, <= 4 ulp for
pow over the barometric range — so `det` results stand in for what the Rust reference gets from libm."""
import ctypes as C
import math

import mpmath
import numpy as np
import pytest

import cbuild


@pytest.fixture(scope="module")
def dm():
    return C.CDLL(cbuild.dm_export())


def call1(dm, name, x):
    x = np.ascontiguousarray(x, dtype=np.float64)
    y = np.empty_like(x)
    getattr(dm, "t_" + name)(C.c_void_p(x.ctypes.data), C.c_void_p(y.ctypes.data), C.c_size_t(x.size))
    return y


def call2(dm, name, a, b):
    a, b = np.ascontiguousarray(a, dtype=np.float64), np.ascontiguousarray(b, dtype=np.float64)
    y = np.empty_like(a)
    getattr(dm, "t_" + name)(C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data), C.c_void_p(y.ctypes.data), C.c_size_t(a.size))
    return y


def ulps(got, want):
    want = np.asarray(want, dtype=np.float64)
    return np.abs(got - want) / np.spacing(np.abs(want))


RANGES = {
    "sin": [(-0.8, 0.8), (-7, 7), (-1000, 1000)], "cos": [(-0.8, 0.8), (-7, 7), (-1000, 1000)], "tan": [(-1.5, 1.5)],
    "atan": [(-0.5, 0.5), (-3, 3), (-1e6, 1e6)], "asin": [(-0.5, 0.5), (-1, 1), (0.97, 1)], "exp": [(-3, 3), (-700, 700)],
    "log": [(0.5, 2), (1e-300, 1e300)],
}
LIMIT = {"tan": 2.0}


@pytest.mark.parametrize("name", sorted(RANGES))
def test_against_libm(dm, name):
    rng = np.random.default_rng(11)
    for lo, hi in RANGES[name]:
        x = rng.uniform(lo, hi, 400_000)
        got = call1(dm, name, x)
        want = getattr(np, {"asin": "arcsin", "atan": "arctan"}.get(name, name))(x)
        assert ulps(got, want).max() <= LIMIT.get(name, 1.0), (name, lo, hi)


def test_atan2_and_pow_against_libm(dm):
    rng = np.random.default_rng(12)
    y, x = rng.uniform(-10, 10, 400_000), rng.uniform(-10, 10, 400_000)
    assert ulps(call2(dm, "atan2", y, x), np.arctan2(y, x)).max() <= 1.0
    base, ex = rng.uniform(0.7, 1.3, 400_000), rng.uniform(-6, 6, 400_000)  # (T/Tb)^(-g M / R L): |y ln x| < 2.2
    assert ulps(call2(dm, "pow", base, ex), np.power(base, ex)).max() <= 4.0


def test_against_mpmath(dm):
    mpmath.mp.prec = 200
    rng = np.random.default_rng(13)
    fns = {"sin": mpmath.sin, "cos": mpmath.cos, "atan": mpmath.atan, "asin": mpmath.asin, "exp": mpmath.exp, "log": mpmath.log}
    for name, f in fns.items():
        lo, hi = RANGES[name][1] if len(RANGES[name]) > 1 else RANGES[name][0]
        x = rng.uniform(lo, hi, 300)
        got = call1(dm, name, x)
        for xi, gi in zip(x, got):
            exact = f(mpmath.mpf(float(xi)))
            err = abs(mpmath.mpf(float(gi)) - exact) / mpmath.mpf(float(np.spacing(abs(float(exact)))))
            assert err <= 1.0, (name, xi, float(err))


def test_special_values(dm):
    assert call2(dm, "atan2", [0.0, -0.0, 1.0, -1.0, 0.0], [-1.0, -1.0, 0.0, -0.0, 1.0]).tolist() == [math.pi, -math.pi, math.pi / 2, -math.pi / 2, 0.0]
    a = call1(dm, "asin", [1.0, -1.0, 1.0000001, 0.0])
    assert a[0] == math.pi / 2 and a[1] == -math.pi / 2 and math.isnan(a[2]) and a[3] == 0.0
    s = call1(dm, "sin", [0.0, np.inf, np.nan])
    assert s[0] == 0.0 and math.isnan(s[1]) and math.isnan(s[2])
    assert call1(dm, "cos", [0.0])[0] == 1.0
    assert call1(dm, "exp", [0.0, -1000.0, 1000.0]).tolist() == [1.0, 0.0, np.inf]
    assert call1(dm, "log", [1.0])[0] == 0.0
    x = np.random.default_rng(5).uniform(0, 1e6, 100000)
    assert np.array_equal(call1(dm, "sqrt", x), np.sqrt(x))


def test_division_by_tabulated_reciprocal(dm):
    """dm_div_r_seq(a, b, RN(1/b)) — what the GPU uses for T/T_b and dn/(2 eps) — equals the IEEE quotient: the layer base
    temperatures of US-76, 2 eps, and random denominators, 2e6 numerators each over the magnitudes the path sees."""
    rng = np.random.default_rng(14)
    for b in (288.15, 216.65, 228.65, 270.65, 214.65, 0.02, 6371000.0, 1.0, 3.0, 0.1):
        for scale in (1.0, 1e-9, 1e6):
            a = rng.uniform(-400.0, 400.0, 2_000_000) * scale
            bb = np.full_like(a, b)
            assert np.array_equal(call2(dm, "div_r_seq", a, bb), a / bb), (b, scale)
    a, b = rng.uniform(-1e3, 1e3, 4_000_000), rng.uniform(1e-3, 1e3, 4_000_000) * rng.choice([-1.0, 1.0], 4_000_000)
    assert np.array_equal(call2(dm, "div_r_seq", a, b), a / b)
    a, b = np.exp(rng.uniform(-200, 200, 2_000_000)), np.exp(rng.uniform(-200, 200, 2_000_000))
    assert np.array_equal(call2(dm, "div_r_seq", a, b), a / b)
    assert np.array_equal(call2(dm, "div_r_seq", np.zeros(3), np.array([288.15, 0.02, 5.0])), np.zeros(3))

"""The deterministic elementary functions ON THE GPU, element by element, against the host build of the same header.

The bit-exactness of the whole device path rests on two claims: (1) csrc/detmath.h evaluates to the same bits
on gfx950 as on x86-64, and (2) the three GPU-only shortcut sequences dm_div / dm_div_r / dm_sqrt_inrange return the bits of
IEEE division and square root for operands inside their documented range (finite, |binary exponent| < 500 for the operands,
the quotient and the reciprocal).  Round 1 verified both only indirectly, through whole frames.  Here every function runs on
1e7 operands through atmrt_math_probe (the same device code the marching kernels inline) and is compared bit for bit with
tests/csrc/dm_export.c, where the three shortcuts ARE the plain IEEE operations; the operands include the ends of the claimed
range, and the behaviour outside it is recorded."""
import ctypes as C

import numpy as np
import pytest

import cbuild
from util import bits

pytestmark = pytest.mark.gpu

N = 10_000_000
OPS = dict(DIV=0, DIV_R=1, SQRT_INRANGE=2, EXP=3, LOG=4, POW=5, SINCOS=6, ASIN=7, ATAN2=8, IEEE_DIV=9, IEEE_SQRT=10, ATAN=11,
           TAN=12, POW3=13, DIV3=14, DIV3_SEEDED=15, DIV3_SEED_Z=16, DIV_SEED_N=17, POW3_SHARED=18)


@pytest.fixture(scope="module")
def host():
    return C.CDLL(cbuild.dm_export())


def gpu(ctx, op, a, b=None, two=False):
    a = np.ascontiguousarray(a, dtype=np.float64)
    b = None if b is None else np.ascontiguousarray(b, dtype=np.float64)
    o0, o1 = np.empty_like(a), (np.empty_like(a) if two else None)
    ctx.check(ctx.lib.atmrt_math_probe(ctx.handle, OPS[op], a.size, a.ctypes.data, None if b is None else b.ctypes.data,
                                       o0.ctypes.data, None if o1 is None else o1.ctypes.data))
    return (o0, o1) if two else o0


def cpu(host, name, *arrs, outs=1):
    arrs = [np.ascontiguousarray(x, dtype=np.float64) for x in arrs]
    res = [np.empty_like(arrs[0]) for _ in range(outs)]
    fn = getattr(host, "t_" + name)
    fn.restype = None
    fn(*[C.c_void_p(x.ctypes.data) for x in arrs + res], C.c_size_t(arrs[0].size))
    return res[0] if outs == 1 else res


def same(g, w, what):
    gb, wb = bits(g), bits(w)
    nan = np.isnan(g) & np.isnan(w)  # any NaN payload is a NaN
    bad = np.flatnonzero((gb != wb) & ~nan)
    assert bad.size == 0, f"{what}: {bad.size} of {g.size} differ, e.g. index {bad[:3]}: gpu {g[bad[:3]]!r} host {w[bad[:3]]!r}"


def mantissa(rng, n):
    return rng.uniform(1.0, 2.0, n) * rng.choice([-1.0, 1.0], n)


def in_range_pairs(rng, n, lim=499):
    """(a, b) with the binary exponents of a, b, a/b and 1/b all within +-lim, a quarter of them AT the limits."""
    eb = rng.integers(-lim + 1, lim, n)
    eq = rng.integers(-lim + 1, lim, n)  # exponent of the quotient (roughly: the mantissas add at most one)
    edge = rng.uniform(size=n) < 0.25
    eb = np.where(edge, rng.choice([-lim + 1, lim - 1], n), eb)
    eq = np.where(edge & (rng.uniform(size=n) < 0.5), rng.choice([-lim + 2, lim - 2], n), eq)
    ea = np.clip(eb + eq, -lim + 1, lim - 1)
    b = np.ldexp(mantissa(rng, n), eb)
    a = np.ldexp(mantissa(rng, n), ea)
    q = np.abs(a / b)
    ok = (q > 2.0 ** -lim) & (q < 2.0 ** lim)
    a = np.where(ok, a, b * 1.5)
    a[:1000] = 0.0  # a may be +0
    return a, b


def test_division_and_sqrt_are_ieee_on_gfx950(gpu_ctx, host):
    """The compiler's own `/` and sqrt on the GPU are correctly rounded for ANY finite operands (incl. subnormal results)."""
    rng = np.random.default_rng(11)
    a = np.ldexp(mantissa(rng, N), rng.integers(-1060, 1023, N))
    b = np.ldexp(mantissa(rng, N), rng.integers(-1060, 1023, N))
    with np.errstate(all="ignore"):
        same(gpu(gpu_ctx, "IEEE_DIV", a, b), a / b, "a / b")
        x = np.abs(a)
        same(gpu(gpu_ctx, "IEEE_SQRT", x), np.sqrt(x), "sqrt")


def test_dm_div_is_ieee_division_in_range(gpu_ctx, host):
    rng = np.random.default_rng(12)
    a, b = in_range_pairs(rng, N)
    want = a / b
    same(gpu(gpu_ctx, "DIV", a, b), want, "dm_div")
    same(cpu(host, "div", a, b), want, "host dm_div")
    # operands as the march produces them: temperatures, pressures, radii, refractive indices
    a2 = rng.uniform(-1e8, 1e8, N)
    b2 = rng.choice([-1.0, 1.0], N) * 10.0 ** rng.uniform(-12, 12, N)
    same(gpu(gpu_ctx, "DIV", a2, b2), a2 / b2, "dm_div (march-like operands)")


def test_dm_div_r_is_ieee_division_in_range(gpu_ctx, host):
    rng = np.random.default_rng(13)
    a, b = in_range_pairs(rng, N)
    same(gpu(gpu_ctx, "DIV_R", a, b), a / b, "dm_div_r")
    # the two call sites: T / T_b with the tabulated 1/T_b, and (n2 - n1) / 0.02
    t = rng.uniform(150.0, 330.0, N)
    tb = rng.choice([288.15, 216.65, 228.65, 270.65, 214.65, 186.946, 301.3], N)
    same(gpu(gpu_ctx, "DIV_R", t, tb), t / tb, "T / T_b")
    dn = rng.uniform(-1e-6, 1e-6, N) * 10.0 ** rng.uniform(-6, 0, N)
    same(gpu(gpu_ctx, "DIV_R", dn, np.full(N, 0.02)), dn / 0.02, "dn / (2 eps)")


def test_dm_div3_seeded_reciprocals_give_ieee_quotients(gpu_ctx, host):
    """dm_div3 (the two division sites of the three n(h) evaluations of one ODE right-hand side): the outer divisors' reciprocals
    are refined from the centre divisor's instead of from v_rcp_f64; the quotients must be IEEE's all the same."""
    rng = np.random.default_rng(16)
    a, b = in_range_pairs(rng, N, lim=480)
    with np.errstate(all="ignore"):
        q1, q2 = gpu(gpu_ctx, "DIV3", a, b, two=True)
        same(q1, a / (b * 0.99999976158142090), "dm_div3 outer divisor below")
        same(q2, a / (b * 1.00000047683715820), "dm_div3 outer divisor above")
    h1, h2 = cpu(host, "div3", a, b, outs=2)
    same(q1, h1, "dm_div3 vs host")
    same(q2, h2, "dm_div3 vs host")
    # the call sites' operands: pressures over temperatures, and k (p/T) over Z
    p = rng.uniform(100.0, 115000.0, N)
    t = rng.uniform(150.0, 330.0, N)
    q1, q2 = gpu(gpu_ctx, "DIV3", p, t, two=True)
    same(q1, p / (t * 0.99999976158142090), "p / T")
    same(q2, p / (t * 1.00000047683715820), "p / T")
    kz = rng.uniform(1e-7, 3e-4, N)
    z = 1.0 - rng.uniform(0.0, 1.2e-3, N)
    q1, q2 = gpu(gpu_ctx, "DIV3", kz, z, two=True)
    same(q1, kz / (z * 0.99999976158142090), "k pt / Z")
    same(q2, kz / (z * 1.00000047683715820), "k pt / Z")
    # wavefronts in which some lanes hold NaN / zero divisors: the vote fails and everything goes through dm_div
    b2 = np.where(rng.uniform(size=N) < 0.01, rng.choice([np.nan, np.inf], N), t)
    with np.errstate(all="ignore"):
        q1, q2 = gpu(gpu_ctx, "DIV3", p, b2, two=True)
        ok = np.isfinite(b2)
        same(q1[ok], (p / (b2 * 0.99999976158142090))[ok], "dm_div3 beside NaN lanes")


def test_seeded_divisions_of_tight_segments_give_ieee_quotients(gpu_ctx, host):
    """Round 4: on a TIGHT atmosphere segment (atm_certify) the bounds dm_div3 votes on are certified, so its seeded reciprocals
    run without the vote (dm_div3_seeded), the reciprocal of Z = 1 - small is seeded by 2 - Z and that of n = 1 + q by 1 - q
    (dm_div_seeded) in place of v_rcp_f64.  Every quotient must be the IEEE quotient: 1e7 operands per site from the call sites'
    own populations, up to the certified bounds' edges (|1 - Z|, q up to 2^-10.5 exactly; divisors 2^-22 apart) and, for the
    seeds, beyond them (2^-10: the seed error the unseeded dm_div3 already votes for)."""
    rng = np.random.default_rng(41)
    lo, hi = 0.99999976158142090, 1.00000023841857910  # 1 -+ 2^-22
    # p / T: no seed for the centre, no vote for the outer two
    p = rng.uniform(100.0, 115000.0, N)
    t = rng.uniform(150.0, 330.0, N)
    q1, q2 = gpu(gpu_ctx, "DIV3_SEEDED", p, t, two=True)
    same(q1, p / (t * lo), "p / T below")
    same(q2, p / (t * hi), "p / T above")
    h1, h2 = cpu(host, "div3_seeded", p, t, outs=2)
    same(q1, h1, "dm_div3_seeded vs host")
    same(q2, h2, "dm_div3_seeded vs host")
    a, b = in_range_pairs(rng, N, lim=480)
    with np.errstate(all="ignore"):
        q1, q2 = gpu(gpu_ctx, "DIV3_SEEDED", a, b, two=True)
        same(q1, a / (b * lo), "dm_div3_seeded over the exponent range, below")
        same(q2, a / (b * hi), "dm_div3_seeded over the exponent range, above")
    # K (p / T) / Z with Z = 1 - small: centre seeded by 2 - Z
    kz = rng.uniform(1e-7, 3e-4, N)
    z = 1.0 - rng.uniform(-6.9053396600248786e-04, 6.9053396600248786e-04, N)
    z[: N // 8] = 1.0 - rng.choice([6.9053396600248786e-04, -6.9053396600248786e-04, 0.0, 2.0 ** -30, 4.0e-4, 2.0 ** -10, -(2.0 ** -10)], N // 8)
    q0, q2 = gpu(gpu_ctx, "DIV3_SEED_Z", kz, z, two=True)
    same(q0, kz / z, "K pt / Z, centre seeded by 2 - Z")
    same(q2, kz / (z * hi), "K pt / Z above")
    h0, h2 = cpu(host, "div3_seed_z", kz, z, outs=2)
    same(q0, h0, "dm_div3_seeded(seed) vs host")
    same(q2, h2, "dm_div3_seeded(seed) vs host")
    # X / n with n = 1 + q, seeded by 1 - q: the numerators of the ODE's right-hand side span many binades
    q = rng.uniform(0.0, 6.9053396600248786e-04, N)
    q[: N // 8] = rng.choice([6.9053396600248786e-04, 0.0, 2.0 ** -40, 2.8e-4, 2.0 ** -10], N // 8)
    x = mantissa(rng, N) * 2.0 ** rng.integers(-60, 60, N)
    got = gpu(gpu_ctx, "DIV_SEED_N", x, q)
    same(got, x / (1.0 + q), "X / n seeded by 1 - (n - 1)")
    same(got, cpu(host, "div_seed_n", x, q), "dm_div_seeded vs host")


def test_shared_row_pow_of_tight_segments_is_the_plain_pow(gpu_ctx, host):
    """Round 4: on a TIGHT segment the three pow evaluations of one right-hand side share one log table row and one exp table entry
    where the centre argument is not at a table edge (csrc/detmath.h dm_log3_core_pow_shared / dm_exp3_main_shared; wave votes, else
    the plain calls).  The values must be dm_pow's bit for bit: the barometric population, bases AT the edges of the log table's
    intervals (centre +- one unit of the high word, where a neighbour's row differs) and exponent products at the half-integers of
    exp's reduction — in wavefronts of their own and mixed into ordinary wavefronts."""
    rng = np.random.default_rng(43)
    x = rng.uniform(0.72, 1.39, N)
    y = rng.choice([5.2558761132785179, -17.08, 34.163, -3.4e-4 * 288.0, 1.0, -11.388], N)
    # bases on the edges of the 128 table intervals: high word with (hi + 0x1000) & 0x1fff in {0x1fff, 0, 1}, any low word
    edge = rng.integers(0, N, N // 6)
    hi = (0x3fe6a000 + rng.integers(0, 128, edge.size) * 0x2000 + rng.choice([-0x1001, -0x1000, -0xfff, 0xfff, 0x1000], edge.size)).astype(np.uint64)
    lo = rng.integers(0, 2 ** 32, edge.size).astype(np.uint64)
    lo[::3] = rng.choice(np.array([0, 1, 2 ** 32 - 1], dtype=np.uint64), lo[::3].size)
    x[edge] = ((hi << np.uint64(32)) | lo).view(np.float64)
    x[: N // 8] = np.sort(x[: N // 8])  # wavefronts of neighbours: whole wavefronts on one row, some of them all on an edge
    # exponent products at the half-integers of exp's reduction: y log x = (k + 1/2) ln2 / 128 up to rounding
    half = rng.integers(N // 2, N, N // 6)
    kk = rng.integers(-60, 60, half.size)
    with np.errstate(all="ignore"):
        x[half] = np.exp((kk + 0.5) * np.log(2.0) / 128.0 / y[half]) * (1.0 + rng.choice([0.0, 1e-9, -1e-9, 4e-7, -4e-7], half.size))
    x = np.clip(x, 0.5, 1.9)
    g0, g12 = gpu(gpu_ctx, "POW3_SHARED", x, y, two=True)
    h0, h12 = cpu(host, "pow3_shared", x, y, outs=2)
    same(g0, h0, "shared-row pow, centre")
    same(g12, h12, "shared-row pow, outer points")
    same(g0, gpu(gpu_ctx, "POW", x, y), "shared-row pow against the plain device pow")


def test_dm_sqrt_inrange_is_ieee_sqrt_in_range(gpu_ctx, host):
    rng = np.random.default_rng(14)
    e = rng.integers(-499, 500, N)
    e[: N // 4] = rng.choice([-499, -498, 498, 499], N // 4)
    x = np.ldexp(rng.uniform(1.0, 2.0, N), e)
    x[:100] = np.ldexp(1.0, np.arange(-50, 50)).astype(np.float64)  # exact powers of two / perfect squares
    same(gpu(gpu_ctx, "SQRT_INRANGE", x), np.sqrt(x), "dm_sqrt_inrange")
    d = rng.uniform(1.0, 1e5, N)
    x2 = d * d + rng.uniform(-1e3, 1e3, N) ** 2  # calc_dist's radicand
    same(gpu(gpu_ctx, "SQRT_INRANGE", x2), np.sqrt(x2), "dm_sqrt_inrange (calc_dist operands)")


def test_shortcuts_outside_their_range_are_not_ieee(gpu_ctx, host):
    """What the range restriction buys: outside it the sequences are NOT the IEEE operations (no scaling, no fix-up), which is
    why every call site is argued to stay inside and why out-of-envelope frames are tested separately
    (test_out_of_envelope_rays).  Recorded, not relied upon."""
    rng = np.random.default_rng(15)
    n = 1_000_000
    a = np.ldexp(mantissa(rng, n), rng.integers(900, 1023, n))
    b = np.ldexp(mantissa(rng, n), rng.integers(-1022, -900, n))
    with np.errstate(all="ignore"):
        want = a / b  # overflows to inf
        got = gpu(gpu_ctx, "DIV", a, b)
        frac = float(np.mean((bits(got) != bits(want)) & ~(np.isnan(got) & np.isnan(want))))
        print(f"dm_div with overflowing quotients: {100 * frac:.1f} % of the results differ from IEEE")
        z = gpu(gpu_ctx, "DIV", np.array([1.0, 0.0, np.inf]), np.array([0.0, 0.0, 2.0]))
        print("dm_div(1, 0), (0, 0), (inf, 2) =", z)
        s = gpu(gpu_ctx, "SQRT_INRANGE", np.array([0.0, np.inf, 4e-320, 1e-310]))
        print("dm_sqrt_inrange(0, inf, subnormals) =", s)
    assert 0.0 <= frac <= 1.0


@pytest.mark.parametrize("name", ["EXP", "LOG", "ASIN", "ATAN", "TAN"])
def test_one_argument_functions_match_the_host_build(gpu_ctx, host, name):
    rng = np.random.default_rng(20 + OPS[name])
    if name == "EXP":
        x = np.concatenate([rng.uniform(-700, 700, N // 2), rng.normal(0, 3, N // 4), rng.uniform(-800, 800, N // 8),
                            rng.uniform(-1e-3, 1e-3, N // 8), [0.0, -0.0, 709.78, 709.79, -745.0, -746.0, np.inf, -np.inf, np.nan]])
    elif name == "LOG":
        x = np.concatenate([np.ldexp(rng.uniform(1, 2, N // 2), rng.integers(-1074, 1023, N // 2)), rng.uniform(0.5, 1.5, N // 4),
                            1.0 + rng.normal(0, 1e-4, N // 4), [0.0, -0.0, -1.0, np.inf, np.nan, 5e-324, 2.2250738585072014e-308]])
    elif name == "ASIN":
        x = np.concatenate([rng.uniform(-1, 1, N // 2), np.sin(rng.uniform(-1.6, 1.6, N // 4)), 1 - 10.0 ** rng.uniform(-16, 0, N // 8),
                            rng.uniform(-1e-7, 1e-7, N // 8), [1.0, -1.0, 1.0000001, 0.5, 0.975, 0.0, np.nan]])
    elif name == "ATAN":
        x = np.concatenate([rng.uniform(-3, 3, N // 2), np.ldexp(mantissa(rng, N // 2), rng.integers(-60, 70, N // 2)),
                            [0.4375, 0.6875, 1.1875, 2.4375, np.inf, -np.inf, np.nan, 0.0]])
    else:
        x = np.concatenate([rng.uniform(-1.5, 1.5, N // 2), rng.uniform(-1e4, 1e4, N // 2)])
    with np.errstate(all="ignore"):
        same(gpu(gpu_ctx, name, x), cpu(host, name.lower(), x), name)


def test_sincos_matches_the_host_build(gpu_ctx, host):
    rng = np.random.default_rng(31)
    # (a) every lane within pi/4: the wave-vote shortcut that skips the reduction; (b) mixed wavefronts; (c) large arguments
    for label, x in (("|x| <= pi/4", rng.uniform(-0.78539816339744828, 0.78539816339744828, N)),
                     ("mixed", np.where(rng.uniform(size=N) < 0.9, rng.uniform(-0.7, 0.7, N), rng.uniform(-50, 50, N))),
                     ("large", np.concatenate([rng.uniform(-1e6, 1e6, N - 6), [0.0, -0.0, np.pi / 4, np.inf, np.nan, 1.6e6]]))):
        with np.errstate(all="ignore"):
            s, c = gpu(gpu_ctx, "SINCOS", x, two=True)
            hs, hc = cpu(host, "sincos", x, outs=2)
        same(s, hs, "sin " + label)
        same(c, hc, "cos " + label)


def test_atan2_and_pow_match_the_host_build(gpu_ctx, host):
    rng = np.random.default_rng(32)
    y = np.ldexp(mantissa(rng, N), rng.integers(-80, 80, N))
    x = np.ldexp(mantissa(rng, N), rng.integers(-80, 80, N))
    y[:8] = [0.0, -0.0, 0.0, 1.0, -1.0, np.inf, np.inf, np.nan]
    x[:8] = [1.0, -1.0, 0.0, 0.0, 0.0, np.inf, -np.inf, 1.0]
    with np.errstate(all="ignore"):
        same(gpu(gpu_ctx, "ATAN2", y, x), cpu(host, "atan2", y, x), "atan2")
    # barometric formula: base T / T_b in (0.2, 2), exponent -g M / (R lapse) for lapses of 0.5 .. 50 K/km
    base = rng.uniform(0.2, 2.0, N)
    expo = rng.choice([-1.0, 1.0], N) * 0.0341631947 / 10.0 ** rng.uniform(-3.3, -1.3, N)
    same(gpu(gpu_ctx, "POW", base, expo), cpu(host, "pow", base, expo), "pow")
    p0, p1 = gpu(gpu_ctx, "POW3", base, expo, two=True)
    h0, h1 = cpu(host, "pow3", base, expo, outs=2)
    same(p0, h0, "pow3 centre")
    same(p1, h1, "pow3 outer points")
    # wavefronts in which SOME lanes leave the main range of log / exp (the voted fast path must not be taken for them)
    base2 = np.where(rng.uniform(size=N) < 0.02, rng.choice([0.0, -1.0, 1e-310, np.inf], N), base)
    expo2 = np.where(rng.uniform(size=N) < 0.02, 1e4, expo)
    with np.errstate(all="ignore"):
        p0, p1 = gpu(gpu_ctx, "POW3", base2, expo2, two=True)
        h0, h1 = cpu(host, "pow3", base2, expo2, outs=2)
    same(p0, h0, "pow3 centre, guarded lanes")
    same(p1, h1, "pow3 outer points, guarded lanes")

"""The C-ABI library loads, exports every symbol include/atmrt.h declares, agrees on struct sizes with the
ctypes mirror, and refuses to run without a GPU (no CPU fallback).  No compute calls here."""
import ctypes as C
import os
import re

import pytest

from atm_raytracer_amd import _abi, _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    return _lib.load()


def test_every_declared_symbol_is_exported(lib):
    header = open(os.path.join(ROOT, "include", "atmrt.h")).read()
    declared = set(re.findall(r"\b(atmrt_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    assert declared == set(_lib.EXPORTED), declared ^ set(_lib.EXPORTED)
    for name in declared:
        assert hasattr(lib, name), name


def test_struct_sizes_match_the_header(lib):
    for which, st in enumerate((_abi.Params, _abi.Atmosphere, _abi.Object, _abi.Result, _abi.DevicePlanes, _abi.EarthModel,
                                _abi.Position, _abi.Frame, _abi.FrameStats, _abi.Timings, _abi.Coloring, _abi.DeviceHits, _abi.CommTimings,
                                _abi.TempFunction)):
        assert lib.atmrt_abi_sizeof(which) == C.sizeof(st), st.__name__


def test_defaults_match_the_reference(lib):
    p = _abi.Params()
    lib.atmrt_params_default(C.byref(p))
    assert (p.width, p.height, p.frame.fov, p.frame.max_distance) == (640, 480, 30.0, 150_000.0)  # params.rs:419-425,156-162
    assert (p.simulation_step, p.wavelength, p.terrain_alpha, p.generator) == (50.0, 530e-9, 1.0, 0)  # :473-479,76-78,427-429
    assert (p.earth.kind, p.earth.radius) == (_abi.EARTH_KINDS["Spherical"], 6_371_000.0)  # :467-471
    assert (p.position.altitude_kind, p.position.altitude) == (_abi.ALT_RELATIVE, 1.0)  # :42-44
    a = _abi.Atmosphere()
    lib.atmrt_atmosphere_us76(C.byref(a))
    assert a.n_functions == 7 and a.pressure == 101325.0 and a.temperature == 288.15 and a.functions[0].gradient == -0.0065
    assert a.has_temperature_fixed_point == 1 and a.functions[6].altitude == 71000.0 and lib.atmrt_abi_version() == 5


def test_build_info_names_the_sources_and_the_required_flags(lib):
    """The library says what it was built from: the hash equals the tree's (a stale .so would be caught here), and the units whose
    kernels call device functions were compiled without interprocedural register allocation (profiles/r03/ipra/README.md)."""
    info = _lib.build_info()
    assert info["source_hash"] == _lib.source_hash(), "libatmrt.so was not built from this tree: run make -C atm-raytracer_amd/csrc"
    assert "-enable-ipra=0" in info["calling_units"] and "-disable-machine-licm" in info["march_units"] and info["arch"] == "gfx950"
    # round 4 (profiles/r04/ipra/README.md part 3): the units under register pressure are built without the splitting that asks for spill
    # code at block heads (hipcc 7.2 can put it ahead of the block's exec restore); make check-isa verifies the result
    split_off = ("-grow-region-complexity-budget=0", "-vgpr-regalloc=basic")  # region splitting off, or the allocator that never splits
    assert any(f in info["trace_units"] for f in split_off) and any(f in info["march_units"] for f in split_off)
    assert "-ffp-contract=off" in info["all"]


def test_no_cpu_fallback(lib):
    import torch
    if torch.cuda.device_count() > 0:
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    assert lib.atmrt_ctx_create(C.byref(h), 0) == _abi.ERR_NO_DEVICE
    assert b"no CPU path" in lib.atmrt_last_error(None)
    devices = (C.c_int32 * 2)(0, 1)
    assert lib.atmrt_ctx_create_multi(C.byref(h), devices, 2) == _abi.ERR_NO_DEVICE and not h.value
    from atm_raytracer_amd import generators
    with pytest.raises(_lib.AtmrtError):
        generators.Context(0)


def test_march_plan_of_frames_and_tiles(lib):
    """Host logic of the time-sliced march (csrc/atmrt_kernels.h march_slice_layout): which launches take it and the sizes it reserves.
    The FIFO must hold one entry per group and slice a ray can survive; the grid bound of its second kernel (alive groups x
    slices) relies on `slices_after` covering every step count the march can reach (samples + 1 steps, the first slice excluded)."""
    out = (C.c_uint64 * 6)()

    def plan(w, h, samples, objects=0):
        assert lib.atmrt_debug_march_plan(w, h, samples, objects, out) == 0
        return list(out)

    slice_steps = plan(8, 8, 10)[5]
    assert slice_steps == 128
    assert plan(4096, 2048, 2000)[0] == 0, "the whole headline frame (32768 workgroups) marches unsliced"
    sliced, groups, cap, nbytes, after, _ = plan(512, 2048, 2000)  # a column tile of 8 GPUs
    assert sliced == 1 and groups == 512 * 2048 // 64
    assert after * slice_steps >= 2000 + 2 and (after - 1) * slice_steps < 2000 + 2 + slice_steps
    assert cap == groups * after
    assert nbytes == 512 * 2048 * (7 * 8 + 3 * 4 + 128) + 64 + 4 * cap
    assert plan(2048, 2048, 2000)[0] == 1 and plan(2049, 2048, 2000)[0] == 0  # 16384 workgroups of 256 pixels is the limit
    assert plan(512, 2048, 2000, objects=3)[0] == 0, "scenes with objects keep the small-launch march (sliced only when forced)"
    assert plan(64, 64, slice_steps - 2)[0] == 0 and plan(64, 64, slice_steps - 1)[0] == 1  # rays of one slice are not sliced
    ragged = plan(150, 61, 2308)
    assert ragged[1] == (150 * 61 + 63) // 64 and ragged[3] == ragged[1] * 64 * 196 + 64 + 4 * ragged[2]
    assert lib.atmrt_debug_march_plan(-1, 1, 1, 0, out) != 0


def test_tiles_rebalance_is_a_pure_deterministic_rule(lib):
    """atmrt_tiles_rebalance (what every rank evaluates on the same gathered tile times after a frame): boundaries move towards
    the slow tiles, the result is a valid tiling of the same width, balanced inputs stay, bad inputs are refused, and widths never
    fall below one column — also when one tile is a thousand times slower than the others."""
    def rebalance(width, cols, ms):
        n = len(ms)
        out = (C.c_int32 * (n + 1))()
        rc = lib.atmrt_tiles_rebalance(width, n, (C.c_int32 * (n + 1))(*cols), (C.c_double * n)(*ms), out)
        return rc, list(out)

    W = 4096
    equal = [g * W // 8 for g in range(9)]
    rc, same = rebalance(W, equal, [33.0] * 8)
    assert rc == 0 and same == equal
    # the last two tiles 5 % slower (the headline's low-sky columns): a boundary moves only in steps of 64 columns (a wavefront is 64
    # adjacent pixels: unaligned tiles march 15 % slower), and 64 of 512 columns is more than 5 % — the tiling stays
    rc, cols = rebalance(W, equal, [32.2] * 6 + [33.7, 33.7])
    assert rc == 0 and cols == equal
    # one tile twice as slow as the others: it gives columns away in multiples of 64, and the model's slowest tile gets faster
    ms = [30.0] * 7 + [60.0]
    rc, cols = rebalance(W, equal, ms)
    assert rc == 0 and cols[0] == 0 and cols[-1] == W and all(b > a for a, b in zip(cols, cols[1:]))
    assert all(c % 64 == 0 for c in cols) and cols[8] - cols[7] < 512 < cols[1] - cols[0]
    cost = [sum(m * max(0, min(b, e1) - max(a, e0)) / 512 for m, e0, e1 in zip(ms, equal, equal[1:])) for a, b in zip(cols, cols[1:])]
    assert max(cost) < 0.75 * max(ms), cost
    assert rebalance(W, equal, ms)[1] == cols  # deterministic
    # narrow images (fewer than 128 columns per tile) are cut to the column
    rc, cols = rebalance(400, [0, 100, 200, 300, 400], [10.0, 10.0, 10.0, 20.0])
    assert rc == 0 and cols[0] == 0 and cols[-1] == 400 and cols[4] - cols[3] < 100 and any(c % 64 for c in cols)
    # one pathological tile
    rc, cols = rebalance(16, [0, 2, 4, 6, 8, 10, 12, 14, 16], [1.0] * 7 + [1000.0])
    assert rc == 0 and cols[0] == 0 and cols[-1] == 16 and all(b > a for a, b in zip(cols, cols[1:]))
    rc, cols = rebalance(8, list(range(9)), [1000.0] + [1.0] * 7)
    assert rc == 0 and cols == list(range(9))  # nothing to give: every tile is one column wide
    # refused: a non-positive or non-finite time, a tiling that does not span the width, an empty tile
    assert rebalance(W, equal, [33.0] * 7 + [0.0])[0] != 0
    assert rebalance(W, equal, [33.0] * 7 + [float("nan")])[0] != 0
    assert rebalance(W, equal[:-1] + [W - 1], [33.0] * 8)[0] != 0
    assert rebalance(W, [0, 512, 512] + equal[3:], [33.0] * 8)[0] != 0
    assert lib.atmrt_comm_available() in (0, 1)

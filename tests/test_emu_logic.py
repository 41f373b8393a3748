"""Planner logic of the kernel headers, checked on the CPU.

tests/emu/fot_emu.cpp runs csrc/fot_math.hpp + csrc/fot_setup.hpp (the code the
gfx950 kernels are compiled from) with loops instead of a launch grid.  It is
compared here with the reference's golden vectors, so arithmetic/logic errors
surface without a GPU.  The GPU parity tests proper are tests/test_gpu_*.py.
"""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT, wrap_angle
from integrated_path_planning_amd import _abi
from integrated_path_planning_amd.batch import PackedBatch, PlanRequest
from integrated_path_planning_amd.params import make_params

EMU_DIR = os.path.join(ROOT, "tests", "emu")
EMU_SO = os.path.join(EMU_DIR, "_build", "libfot_emu.so")
CSRC = os.path.join(ROOT, "integrated_path_planning_amd", "csrc")


@pytest.fixture(scope="module")
def emu():
    srcs = [os.path.join(EMU_DIR, "fot_emu.cpp")] + [os.path.join(CSRC, f) for f in
                                                      ("fot_math.hpp", "fot_setup.hpp", "fot_types.h")]
    srcs.append(os.path.join(ROOT, "include", "fot.h"))
    if not os.path.exists(EMU_SO) or os.path.getmtime(EMU_SO) < max(os.path.getmtime(s) for s in srcs):
        os.makedirs(os.path.dirname(EMU_SO), exist_ok=True)
        subprocess.run(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", EMU_SO, srcs[0]],
                       check=True)
    L = C.CDLL(EMU_SO)
    dp = C.POINTER(C.c_double)
    L.emu_plan_batch.argtypes = [C.POINTER(_abi.Params), C.c_int, dp, dp, C.POINTER(_abi.Batch),
                                 C.POINTER(_abi.Result), C.c_int, dp, C.POINTER(C.c_int32),
                                 C.POINTER(C.c_int32), C.c_char_p]
    L.emu_spline.argtypes = [C.c_int, dp, dp, dp]
    return L


def request_from_golden(g):
    m = g.meta
    e = m["ego"]
    return PlanRequest(x=e[0], y=e[1], yaw=e[2], v=e[3], a=e[4], target_speed=m["target_speed"],
                       last_kappa=m["last_kappa"], prev_s=m["prev_s"], overrides=m["overrides"],
                       max_stop_distance=m["max_stop"], static=g.static, dyn=g.dyn, dist=g.dist)


def run_emu(emu, g):
    params = make_params(**g.planner_kwargs())
    pb = PackedBatch([request_from_golden(g)])
    wx = np.ascontiguousarray(g["wx"]); wy = np.ascontiguousarray(g["wy"])
    out = (_abi.Result * 1)()
    cap = 16384
    cost = np.zeros(cap); status = np.zeros(cap, np.int32); keep = np.zeros(cap, np.int32)
    err = C.create_string_buffer(256)
    dp = C.POINTER(C.c_double)
    rc = emu.emu_plan_batch(C.byref(params), len(wx), wx.ctypes.data_as(dp), wy.ctypes.data_as(dp), C.byref(pb.c),
                            out, cap, cost.ctypes.data_as(dp), status.ctypes.data_as(C.POINTER(C.c_int32)),
                            keep.ctypes.data_as(C.POINTER(C.c_int32)), err)
    assert rc == 0, err.value
    n = out[0].n_cand
    return out[0], cost[:n], status[:n], keep[:n]


def test_native_spline_fit(emu, golden):
    wx = np.ascontiguousarray(golden["wx"]); wy = np.ascontiguousarray(golden["wy"])
    n = len(wx)
    out = np.zeros(9 * n)
    dp = C.POINTER(C.c_double)
    assert emu.emu_spline(n, wx.ctypes.data_as(dp), wy.ctypes.data_as(dp), out.ctypes.data_as(dp)) == 0
    out = out.reshape(9, n)
    for f, key in enumerate(["sp_s", "sp_ax", "sp_bx", "sp_cx", "sp_dx", "sp_ay", "sp_by", "sp_cy", "sp_dy"]):
        want = golden[key]
        np.testing.assert_allclose(out[f, :len(want)], want, rtol=1e-10, atol=1e-11, err_msg=key)


def test_kernel_logic_matches_reference(emu, golden):
    r, cost, status, keep = run_emu(emu, golden)
    g = golden
    np.testing.assert_allclose(np.array(r.frenet0[:]), g["frenet0"], rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(r.new_prev_s, float(g["prev_s_after"]), atol=1e-12)
    assert r.n_cand == len(g["cand_cost"])
    np.testing.assert_array_equal(keep, g["cand_keep"])
    np.testing.assert_allclose(cost, g["cand_cost"], rtol=1e-9, atol=1e-9)
    np.testing.assert_array_equal(status, g["cand_status"].astype(np.int32))
    stats = g["stats"]
    for i in range(8):
        assert r.stats[i] == max(int(stats[i]), 0), _abi.STATUS_NAMES[i]
    bi = int(g["best_index"])
    assert r.best_index == bi
    if bi < 0:
        assert r.status == _abi.PLAN_NO_PATH
        return
    assert r.status == _abi.PLAN_OK and r.n_keep == len(g["best_x"])
    np.testing.assert_allclose(r.cost, float(g["best_cost"]), rtol=1e-9)
    for f in _abi.PATH_FIELDS:
        got = np.array(getattr(r, f)[: r.n_keep])
        want = g["best_" + f]
        if f == "yaw":
            np.testing.assert_allclose(wrap_angle(got - want), 0.0, atol=1e-9)
        else:
            np.testing.assert_allclose(got, want, rtol=1e-8, atol=1e-8, err_msg=f)
    np.testing.assert_allclose(r.new_last_kappa, float(g["last_kappa_after"]), rtol=1e-8, atol=1e-9)


def test_tile_tables_of_random_lattices(emu):
    """Both cuts of the lattice (per-wave rows / groups of four tiles) for random planner parameters and every
    terminal-speed grid size: consecutive, complete, within the limits the kernels rely on; the handle's choice is one
    of the two with its profile spans filled in."""
    emu.emu_check_tile_tables.argtypes = [C.POINTER(_abi.Params)] + [C.POINTER(C.c_int32)] * 3
    rng = np.random.default_rng(11)
    chosen = []
    for case in range(300):
        dt = float(rng.choice([0.05, 0.1, 0.125, 0.2, 0.25]))
        min_t = float(rng.choice([1.0, 2.0, 3.0, 4.0]))
        max_t = min(min_t + float(rng.choice([0.0, 0.5, 1.0, 2.0])), 63 * dt)
        min_t = min(min_t, max_t)
        kw = dict(dt=dt, min_t=min_t, max_t=max_t, d_road_w=float(rng.choice([0.2, 0.25, 0.5, 1.0, 3.5])),
                  max_road_width=float(rng.uniform(0.5, 7.5)), d_t_s=float(rng.uniform(0.5, 3.0)))
        try:
            params = make_params(**kw)
        except Exception:
            continue
        a, b, c = C.c_int32(0), C.c_int32(0), C.c_int32(0)
        rc = emu.emu_check_tile_tables(C.byref(params), C.byref(a), C.byref(b), C.byref(c))
        if rc == -1:                                     # beyond a FOT_MAX_* limit: not a lattice libfot accepts
            continue
        assert rc == 0, (rc, kw)
        chosen.append(c.value)
    assert len(chosen) > 150 and sum(chosen) > 0                      # (the grouped cut wins on every lattice tried so far)


def test_hypot_and_cube_are_the_references_bit_for_bit(emu):
    """The nearest-point search decides by comparing math.hypot values of probes micrometres apart, positions from
    a + b h + c h**2.0 + d h**3.0 (coordinate_converter.py:253-280, cubic_spline.py:70-71).  hypot_cr of the kernel
    headers must be Python's math.hypot bit for bit (= the correctly rounded hypotenuse), cube_cr the correctly rounded
    cube, spline_xy the reference's operation order with every operation rounded once -- on operands of the magnitudes
    the search sees and on near-ties."""
    import math
    dp = C.POINTER(C.c_double)
    rng = np.random.default_rng(7)
    n = 400_000
    x = np.concatenate([rng.uniform(-200, 200, n), rng.uniform(-2, 2, n), rng.normal(0, 1e-3, n), rng.uniform(-5, 5, n)])
    y = np.concatenate([rng.uniform(-200, 200, n), rng.uniform(-2, 2, n), rng.normal(0, 1e-3, n),
                        rng.uniform(-5, 5, n) * 1e-9])
    # neighbours in the last place: what a tie between two probes looks like
    x = np.concatenate([x, np.nextafter(x[:n], np.inf), x[:n]])
    y = np.concatenate([y, y[:n], np.nextafter(y[:n], -np.inf)])
    out = np.empty_like(x)
    emu.emu_hypot_cr(len(x), x.ctypes.data_as(dp), y.ctypes.data_as(dp), out.ctypes.data_as(dp))
    # (np.hypot -- glibc 2.35's hypot -- is NOT the yardstick: it differs from the correctly rounded value, and from
    #  math.hypot, in about 0.6 % of the calls; the reference's search calls math.hypot)
    from decimal import Decimal, getcontext
    getcontext().prec = 60
    exact = np.array([float((Decimal(float(a)) ** 2 + Decimal(float(b)) ** 2).sqrt()) for a, b in zip(x[:20000], y[:20000])])
    assert np.array_equal(out[:20000], exact), "hypot_cr is not the correctly rounded hypotenuse"
    sel = rng.integers(0, len(x), 300_000)
    want = np.array([math.hypot(float(x[i]), float(y[i])) for i in sel])
    assert np.array_equal(out[sel], want), "hypot_cr differs from Python's math.hypot"
    for a, b, w in ((3.0, 4.0, 5.0), (0.0, -2.5, 2.5), (np.inf, np.nan, np.inf), (1e-200, 1e-200, math.hypot(1e-200, 1e-200))):
        o = np.empty(1)
        emu.emu_hypot_cr(1, np.array([a]).ctypes.data_as(dp), np.array([b]).ctypes.data_as(dp), o.ctypes.data_as(dp))
        assert o[0] == w, (a, b, o[0], w)
    o = np.empty(1)
    emu.emu_hypot_cr(1, np.array([np.nan]).ctypes.data_as(dp), np.array([1.0]).ctypes.data_as(dp), o.ctypes.data_as(dp))
    assert np.isnan(o[0])
    # the cube: correctly rounded.  (NumPy's power on an ARRAY -- what the reference evaluates, cubic_spline.py:70-71 behind
    # np.atleast_1d -- is a SIMD routine that misses the correctly rounded cube in ~5 % of the calls on an AVX512 host and
    # differs from glibc's pow(): the reference's own last bit depends on the NumPy build, so no implementation can be
    # "the reference's" there; the library and the oracle both take the platform-independent value.)
    from fractions import Fraction
    h = np.concatenate([rng.uniform(0, 30, 20000), rng.uniform(0, 1, 20000), rng.uniform(0, 1e-3, 20000)])
    c = np.empty_like(h)
    emu.emu_cube_cr(len(h), h.ctypes.data_as(dp), c.ctypes.data_as(dp))
    exact3 = np.array([float(Fraction(float(v)) ** 3) for v in h])             # (Fraction -> float rounds correctly)
    assert np.array_equal(c, exact3), "cube_cr is not the correctly rounded cube"
    # the probe position, against the reference's expression evaluated by NumPy on the reference's coefficients
    from conftest import Golden
    for name in ("curved_a", "cfg2_s2"):
        g = Golden(name)
        k = len(g["sp_s"])
        pad = lambda a: np.concatenate([a, np.zeros(k - len(a))])
        coef = np.ascontiguousarray(np.concatenate([g["sp_s"], g["sp_ax"], pad(g["sp_bx"]), g["sp_cx"], pad(g["sp_dx"]),
                                                    g["sp_ay"], pad(g["sp_by"]), g["sp_cy"], pad(g["sp_dy"])]))
        s = np.ascontiguousarray(rng.uniform(g["sp_s"][0], g["sp_s"][-1], 20_000))
        px, py = np.empty_like(s), np.empty_like(s)
        emu.emu_spline_xy(k, coef.ctypes.data_as(dp), len(s), s.ctypes.data_as(dp), px.ctypes.data_as(dp), py.ctypes.data_as(dp))
        i = np.clip(np.searchsorted(g["sp_s"], s, side="right") - 1, 0, k - 2)
        dx = s - g["sp_s"][i]
        for got, (a_, b_, c_, d_) in ((px, (g["sp_ax"], g["sp_bx"], g["sp_cx"], g["sp_dx"])),
                                      (py, (g["sp_ay"], g["sp_by"], g["sp_cy"], g["sp_dy"]))):
            cube = np.array([float(Fraction(float(v)) ** 3) for v in dx])
            want = a_[i] + b_[i] * dx + c_[i] * dx ** 2.0 + d_[i] * cube                # cubic_spline.py:70-71
            assert np.array_equal(got, want), name

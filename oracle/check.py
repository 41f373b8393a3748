"""Parity checks of libfot results against the oracle's output.

TEST INFRASTRUCTURE ONLY (like everything under oracle/): used by tests/, by bench.py's parity spot check /
``cpu_baseline`` leg and by ``__graft_entry__.smoke()``.  Nothing under ``integrated_path_planning_amd/`` imports it.
"""
import numpy as np

STATUS_NAMES = ["max_speed_error", "max_accel_error", "max_curvature_error", "max_lat_accel_error",
                "road_bound_error", "collision_error", "ok", "stop_distance_error"]
PATH_FIELDS = ["t", "s", "s_d", "s_dd", "s_ddd", "d", "d_d", "d_dd", "d_ddd", "x", "y", "yaw", "v", "a", "c"]

# north_star tolerance: selected path and cost within 1e-5 of the reference (fp32-level).
# The kernels compute in float64, so the tests hold them to a much tighter bound.
NORTH_STAR_TOL = 1e-5
TIGHT = 1e-8


def wrap_angle(a):
    return (np.asarray(a) + np.pi) % (2 * np.pi) - np.pi


def oracle_plan_for_request(orc, params, spline, req, table=False):
    ego = orc.make_ego(req.x, req.y, req.yaw, req.v, req.a, last_kappa=req.last_kappa, prev_s=req.prev_s)
    return orc.plan(params, spline, ego, req.target_speed, req.overrides, req.max_stop_distance,
                    static=req.static, dyn=req.dyn, dist=req.dist, table=table)


# Nearest point (coordinate_converter.py:202-308).  The refinement compares the distances of three probes micrometres
# apart; at a rounding-level tie the last bit of hypot() and of the probe position picks the probe, and the arc length s0
# ends a refinement step away (0.2 * 2^-k: 3e-6 m at k = 16) -- with everything derived from the start state.  Round 3
# held such instances to the north star's 1e-5.  Round 4: the reference's own last bits there are math.hypot (correctly
# rounded) and NumPy's SIMD power loop for h**3.0 (NOT correctly rounded, and dependent on the NumPy build: ~5 % of the
# cubes differ from glibc's pow on this container's AVX512 host -- tests/test_emu_logic.py), so no implementation can be
# "the reference's" at a tie; the library and the oracle both use the platform-independent values (correctly rounded
# hypot and cube, the reference's operation order: fot_math.hpp hypot_cr / cube_cr / spline_xy, fot_oracle.c py_hypot /
# cube_cr) and must now agree on s0 EXACTLY -- no tolerance, no carve-out.
#
# Curvature at a crawl (coordinate_converter.py:128-158, frenet_planner.py:792-799): d' = d_d / s_d and d'' divide by s_d
# and s_d^2, and s_d itself is what is left of a quartic's terms of metres per second cancelling: a few 1e-16 of absolute
# error in s_d become 1e-13 relative at s_d = 5e-3 and 1e-8 in the curvature (observed: 2.5e-8 at s_d = 1.2e-3, 1.4e-8 at
# 6.8e-3).  A curvature sample with |s_d| < CRAWL_S_DOT is held to CRAWL_C_TOL (a tenth of the north star's 1e-5).
CRAWL_S_DOT = 0.05
CRAWL_C_TOL = 1e-6
tolerance_stats = {"nearest_point_ties": 0, "crawl_curvature_samples": 0, "records": 0}
crawl_labels = []                 # labels of the records that used the crawl allowance (the fuzz prints them)


def assert_record_matches_oracle(rec, want, tol=TIGHT, label=""):
    """fot_result record vs oracle PlanOutput (values at `tol`; the two documented amplifications above at theirs)."""
    tolerance_stats["records"] += 1
    # the arc length of the nearest point: bit for bit (NaN == NaN: an ego given as a Frenet state has none)
    s_got, s_want = float(rec.new_prev_s), float(want.new_prev_s)
    if not (s_got == s_want or (np.isnan(s_got) and np.isnan(s_want))):
        tolerance_stats["nearest_point_ties"] += 1
        raise AssertionError(f"{label} nearest point: s0 {s_got!r} != oracle's {s_want!r} (difference {s_got - s_want:.3e})")
    assert rec.status == want.status, f"{label} status {rec.status} != {want.status}"
    assert rec.best_index == want.best_index, f"{label} best_index {rec.best_index} != {want.best_index}"
    assert rec.n_cand == want.n_cand, label
    if want.stats is not None:
        for k in range(8):
            assert rec.stats[k] == want.stats.get(STATUS_NAMES[k], 0), f"{label} stats[{STATUS_NAMES[k]}]"
    np.testing.assert_allclose(np.array(rec.frenet0[:]), want.frenet0, rtol=tol, atol=tol, err_msg=label)
    np.testing.assert_allclose(rec.new_prev_s, want.new_prev_s, atol=tol, err_msg=label)
    if want.status != 0:
        return
    np.testing.assert_allclose(rec.cost, want.cost, rtol=tol, err_msg=label)
    n = rec.n_keep
    for f in PATH_FIELDS:
        got = np.array(getattr(rec, f)[:n])
        exp = want.path[f]
        assert len(exp) == n, f"{label} len({f})"
        if f == "yaw":
            np.testing.assert_allclose(wrap_angle(got - exp), 0.0, atol=tol, err_msg=f"{label} {f}")
        elif f == "c":
            sd = np.asarray(want.path["s_d"], dtype=float)
            loose = np.where(np.abs(sd) < CRAWL_S_DOT, CRAWL_C_TOL, 0.0)
            err = np.abs(got - exp)
            ok = err <= tol + tol * np.abs(exp) + loose
            n_loose = int(np.sum(ok & (err > tol + tol * np.abs(exp))))
            tolerance_stats["crawl_curvature_samples"] += n_loose
            if n_loose:
                crawl_labels.append((label, float(err.max()), float(sd[int(err.argmax())])))
                import os
                if os.environ.get("FOT_CRAWL_LOG"):                  # (sweeps under pytest-xdist: the workers' reports are not shown)
                    with open(os.environ["FOT_CRAWL_LOG"], "a") as f:
                        f.write(f"{label}\t{err.max():.3e}\t{sd[int(err.argmax())]:.3e}\n")
            assert np.all(ok), f"{label} c: max error {err.max():.3e} at sample {int(err.argmax())} (s_d {sd[int(err.argmax())]:.3e})"
        else:
            np.testing.assert_allclose(got, exp, rtol=tol, atol=tol, err_msg=f"{label} {f}")
    np.testing.assert_allclose(rec.new_last_kappa, want.new_last_kappa, rtol=tol, atol=tol, err_msg=label)

"""``closed_loop.emergency_stop`` against the reference's own expectations of ``IntegratedSimulator._apply_emergency_stop``
(integrated_simulator.py:749-802): the numbers of tests/test_planner_guards.py:34-70 (kinematics) and
tests/test_smooth_braking.py:166-232 (adaptive rate), restated against this build's function."""
import numpy as np
import pytest

from integrated_path_planning_amd.closed_loop import emergency_stop


def one(v=5.0, yaw=0.0, clearance=0.1, x=10.0, y=-3.0, max_accel=2.0, cap=None, dt=0.1):
    nx, ny, nv, na = emergency_stop([x], [y], [yaw], [v], [clearance], dt, max_accel, cap)
    return float(nx[0]), float(ny[0]), float(nv[0]), float(na[0])


def test_position_integrates_during_braking():
    x, y, v, a = one(v=5.0, yaw=0.0)
    np.testing.assert_allclose([x, y, v], [10.5, -3.0, 5.0 - 4.0 * 0.1], atol=1e-12)


def test_position_integrates_along_heading():
    x, y, v, a = one(v=2.0, yaw=np.pi / 2)
    np.testing.assert_allclose([x, y], [10.0, -3.0 + 0.2], atol=1e-9)


def test_braking_distance_over_full_stop():
    x, y, v, a = 10.0, -3.0, 4.0, 0.0
    for _ in range(20):
        x, y, v, a = one(v=v, x=x, y=y)
    assert v == 0.0 and a == 0.0
    assert 1.5 < x - 10.0 < 2.5


def test_inputs_are_not_modified():
    xs, vs = np.array([10.0, 11.0]), np.array([5.0, 3.0])
    keep = xs.copy(), vs.copy()
    emergency_stop(xs, np.zeros(2), np.zeros(2), vs, np.array([0.1, np.inf]), 0.1, 2.0, None)
    np.testing.assert_array_equal(xs, keep[0])
    np.testing.assert_array_equal(vs, keep[1])


def test_config_key_overrides_legacy_rate():
    _, _, v, a = one(clearance=0.1, cap=3.0)
    assert v == pytest.approx(5.0 - 3.0 * 0.1) and a == pytest.approx(-3.0)


def test_none_falls_back_to_twice_max_accel():
    _, _, v, a = one(clearance=0.1, cap=None)
    assert v == pytest.approx(5.0 - 4.0 * 0.1) and a == pytest.approx(-4.0)


def test_adaptive_rate_uses_available_clearance():
    assert one(clearance=5.2, cap=4.0)[3] == pytest.approx(-2.5)          # 25 / (2 * 5.0)


def test_adaptive_rate_floors_at_max_accel_when_room_is_ample():
    assert one(clearance=100.0, cap=4.0)[3] == pytest.approx(-2.0)


def test_nonfinite_clearance_falls_back_to_max_rate():
    assert one(clearance=np.inf, cap=4.0)[3] == pytest.approx(-4.0)
    assert one(clearance=np.inf, cap=None)[3] == pytest.approx(-4.0)


def test_adaptive_rate_saturates_at_cap_when_room_is_gone():
    assert one(clearance=0.2, cap=4.0)[3] == pytest.approx(-4.0)


def test_many_egos_at_once_equal_one_at_a_time():
    rng = np.random.default_rng(0)
    n = 50
    v, yaw = rng.uniform(0, 8, n), rng.uniform(-3, 3, n)
    clr = np.where(rng.random(n) < 0.3, np.inf, rng.uniform(0.0, 30.0, n))
    got = emergency_stop(np.zeros(n), np.zeros(n), yaw, v, clr, 0.1, 2.0, 4.5)
    for i in range(n):
        want = one(v=v[i], yaw=yaw[i], clearance=clr[i], x=0.0, y=0.0, cap=4.5)
        np.testing.assert_allclose([g[i] for g in got], want, rtol=0, atol=0)

"""fail-safe state machine + escalate-and-retry planning cycle.

* the reference's tests/test_state_machine.py restated against the mirror class (same numbers, own code);
* the state machine replayed over the 274 planning cycles of the reference closed loop (CPU);
* SpeculativePlanningCycle (all escalation levels in one launch) replayed over the same cycles on the GPU.
"""
import json
import os
from types import SimpleNamespace

import numpy as np
import pytest

from conftest import GOLDEN_DIR
from integrated_path_planning_amd.state_machine import FailSafeStateMachine, VehicleState

S = VehicleState
BY_ID = [S.NORMAL, S.CAUTION, S.EMERGENCY]


def make_config(**kw):
    """defaults of the reference's SimulationConfig that the state machine reads (config/__init__.py:68-150)"""
    c = SimpleNamespace(ego_target_speed=8.33, ego_max_speed=10.0, ego_max_accel=2.0, ego_max_curvature=1.0,
                        ego_max_lat_accel=3.0, ego_radius=1.0, ped_radius=0.2, ego_footprint="circle",
                        vehicle_length=4.5, vehicle_width=2.0, ego_footprint_n_circles=3,
                        state_machine_safe_distance_caution=2.0, state_machine_safe_distance_emergency=3.0,
                        state_machine_recover_clearance_caution=None, state_machine_recover_clearance_emergency=None,
                        state_machine_trigger_clearance_caution=0.0, state_machine_trigger_time_headway=0.0,
                        state_machine_envelope_decel=0.0, state_machine_envelope_standoff=0.5,
                        state_machine_caution_accel_multiplier=1.5, state_machine_caution_speed_multiplier=0.8,
                        state_machine_emergency_accel_multiplier=3.0, state_machine_emergency_lat_accel_multiplier=2.0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


# ------------------------------------------------------------------ tests/test_state_machine.py restated

def test_basic_transitions():                                                # :15-63
    sm = FailSafeStateMachine(make_config())
    assert sm.current_state == S.NORMAL
    out = sm.update(False, {"clearance": 8.8})
    assert sm.current_state == S.CAUTION and out.state == S.CAUTION and out.constraint_overrides["max_accel"] > 2.0
    out = sm.update(False, {"clearance": 8.8})
    assert sm.current_state == S.EMERGENCY and out.target_speed_override == 0.0
    sm = FailSafeStateMachine(make_config()); sm.current_state = S.CAUTION
    assert sm.update(True, {"clearance": 1.5}).state == S.NORMAL
    sm = FailSafeStateMachine(make_config()); sm.current_state = S.EMERGENCY
    sm.update(True, {"clearance": -0.7}); assert sm.current_state == S.EMERGENCY
    sm.update(True, {"clearance": 3.8}); assert sm.current_state == S.CAUTION


def test_recovery_sequences():                                               # :65-107
    sm = FailSafeStateMachine(make_config())
    sm.update(False, {"clearance": 5.0}); assert sm.current_state == S.CAUTION
    sm.update(True, {"clearance": 5.0}); assert sm.current_state == S.CAUTION
    sm.update(True, {"clearance": 5.0}); assert sm.current_state == S.NORMAL
    sm = FailSafeStateMachine(make_config())
    sm.update(False, {"clearance": 5.0}); sm.update(False, {"clearance": 5.0})
    assert sm.current_state == S.EMERGENCY
    sm.update(True, {"clearance": sm.clearance_emergency - 0.1}); assert sm.current_state == S.EMERGENCY
    sm.update(True, {"clearance": sm.clearance_emergency + 0.5}); assert sm.current_state == S.CAUTION
    sm.update(True, {"clearance": sm.clearance_caution + 0.5}); assert sm.current_state == S.CAUTION
    sm.update(True, {"clearance": sm.clearance_caution + 0.5}); assert sm.current_state == S.NORMAL


def test_thresholds_and_overrides():                                         # :109-143, :253-259
    c = make_config()
    sm = FailSafeStateMachine(c)
    assert sm.clearance_caution == pytest.approx(2.0 - 1.2) and sm.clearance_emergency == pytest.approx(3.0 - 1.2)
    sm.current_state = S.CAUTION
    sm.update(True, {"clearance": 2.0 - 1.2}); assert sm.current_state == S.CAUTION
    sm.current_state = S.CAUTION
    assert "max_curvature" not in sm._get_planner_config().constraint_overrides
    sm.current_state = S.EMERGENCY
    out = sm._get_planner_config()
    assert "max_curvature" not in out.constraint_overrides and out.constraint_overrides["max_lat_accel"] == pytest.approx(6.0)
    sm = FailSafeStateMachine(make_config(state_machine_recover_clearance_caution=1.3,
                                          state_machine_recover_clearance_emergency=1.7))
    assert sm.clearance_caution == pytest.approx(1.3) and sm.clearance_emergency == pytest.approx(1.7)


def test_preventive_trigger():                                               # :145-251
    sm = FailSafeStateMachine(make_config())
    sm.update(True, {"clearance": 0.01}); assert sm.current_state == S.NORMAL
    sm = FailSafeStateMachine(make_config(state_machine_trigger_clearance_caution=1.5,
                                          state_machine_recover_clearance_caution=2.0,
                                          state_machine_recover_clearance_emergency=2.0))
    sm.update(True, {"clearance": 1.0}); assert sm.current_state == S.CAUTION and sm.consecutive_failures == 0
    sm.update(True, {"clearance": 1.8}); assert sm.current_state == S.CAUTION
    sm.update(True, {"clearance": 2.5}); assert sm.current_state == S.NORMAL
    sm.update(True, {"clearance": 2.5}); assert sm.current_state == S.NORMAL
    sm = FailSafeStateMachine(make_config(state_machine_trigger_clearance_caution=1.0, state_machine_trigger_time_headway=0.8,
                                          state_machine_recover_clearance_caution=6.0,
                                          state_machine_recover_clearance_emergency=6.0))
    sm.update(True, {"clearance": 3.0}, ego_speed=1.0); assert sm.current_state == S.NORMAL
    sm.update(True, {"clearance": 3.0}, ego_speed=6.0); assert sm.current_state == S.CAUTION
    sm = FailSafeStateMachine(make_config(state_machine_trigger_clearance_caution=1.0, state_machine_trigger_time_headway=0.25,
                                          state_machine_recover_clearance_caution=2.0,
                                          state_machine_recover_clearance_emergency=2.0))
    sm.current_state = S.CAUTION
    sm.update(True, {"clearance": 2.2}, ego_speed=6.0); assert sm.current_state == S.CAUTION
    sm.update(True, {"clearance": 2.6}, ego_speed=6.0); assert sm.current_state == S.NORMAL
    sm.update(True, {"clearance": 2.6}, ego_speed=6.0); assert sm.current_state == S.NORMAL
    sm.current_state = S.CAUTION
    sm.update(True, {"clearance": 2.2}, ego_speed=2.0); assert sm.current_state == S.NORMAL
    sm = FailSafeStateMachine(make_config(state_machine_trigger_time_headway=0.8))
    sm.update(True, {"clearance": 0.5}, ego_speed=0.0); assert sm.current_state == S.NORMAL
    sm.update(True, {"clearance": 0.5}, ego_speed=2.0); assert sm.current_state == S.CAUTION
    sm = FailSafeStateMachine(make_config(state_machine_trigger_clearance_caution=1.5,
                                          state_machine_recover_clearance_caution=2.0,
                                          state_machine_recover_clearance_emergency=2.0))
    sm.update(True, {"clearance": 1.6}, ego_speed=10.0); assert sm.current_state == S.NORMAL
    sm.update(True, {"clearance": 1.4}, ego_speed=0.0); assert sm.current_state == S.CAUTION


def test_envelope_and_stop_directive():                                      # :261-410
    c = make_config(ego_target_speed=6.0, state_machine_envelope_decel=1.2, state_machine_envelope_standoff=0.5,
                    state_machine_recover_clearance_caution=100.0, state_machine_recover_clearance_emergency=100.0)
    sm = FailSafeStateMachine(c); sm.current_state = S.CAUTION
    sm.update(True, {"clearance": 2.0})
    assert sm._get_planner_config().target_speed_override == pytest.approx((2 * 1.2 * 1.5) ** 0.5)
    sm.update(True, {"clearance": 0.4}); assert sm._get_planner_config().target_speed_override == pytest.approx(0.0)
    sm.update(True, {"clearance": 50.0}); assert sm._get_planner_config().target_speed_override == pytest.approx(4.8)
    c = make_config(ego_target_speed=6.0, state_machine_envelope_decel=1.2, state_machine_envelope_standoff=1.0)
    sm = FailSafeStateMachine(c)
    sm.update(True, {"clearance": 0.4, "clearance_ahead": float("inf")})
    assert sm._get_planner_config().target_speed_override is None
    sm.update(True, {"clearance": 0.4, "clearance_ahead": 3.0})
    assert sm._get_planner_config().target_speed_override == pytest.approx((2 * 1.2 * 2.0) ** 0.5)
    sm = FailSafeStateMachine(c)
    sm.update(True, {"clearance": 3.0}); out = sm._get_planner_config()
    assert out.state == S.NORMAL and out.target_speed_override == pytest.approx((2 * 1.2 * 2.0) ** 0.5)
    sm.update(True, {"clearance": 50.0}); assert sm._get_planner_config().target_speed_override is None
    sm.observe_metrics({"clearance": 3.0})
    assert sm.current_state == S.NORMAL and sm._get_planner_config().target_speed_override == pytest.approx((2 * 1.2 * 2.0) ** 0.5)
    c = make_config(state_machine_envelope_decel=1.2, state_machine_envelope_standoff=1.0,
                    state_machine_recover_clearance_caution=100.0, state_machine_recover_clearance_emergency=100.0)
    sm = FailSafeStateMachine(c); sm.current_state = S.CAUTION
    sm.update(True, {"clearance": 0.8}); out = sm._get_planner_config()
    assert out.target_speed_override == pytest.approx(0.0) and out.max_stop_distance == pytest.approx(0.6)
    sm.update(True, {"clearance": 3.0}); assert sm._get_planner_config().max_stop_distance is None
    sm = FailSafeStateMachine(make_config()); sm.current_state = S.EMERGENCY; sm._last_clearance_ahead = 1.5
    assert sm._get_planner_config().max_stop_distance is None
    sm = FailSafeStateMachine(make_config(state_machine_envelope_decel=1.2)); sm.current_state = S.EMERGENCY
    sm._last_clearance_ahead = 1.5
    assert sm._get_planner_config().max_stop_distance == pytest.approx(1.3)
    sm = FailSafeStateMachine(make_config(ego_target_speed=6.0)); sm.current_state = S.CAUTION
    sm.update(True, {"clearance": 0.1}); assert sm._get_planner_config().target_speed_override == pytest.approx(4.8)
    sm = FailSafeStateMachine(make_config(ego_target_speed=6.0, state_machine_envelope_decel=1.2)); sm.current_state = S.CAUTION
    assert sm._get_planner_config().target_speed_override == pytest.approx(4.8)


# ------------------------------------------------------------------ reference closed loop, per planning cycle

@pytest.fixture(scope="module")
def cycles():
    z = np.load(os.path.join(GOLDEN_DIR, "closed_loop", "scenario01_cv_cycles.npz"), allow_pickle=False)
    d = {k: z[k] for k in z.files}
    d["meta"] = json.loads(str(d["meta"]))
    return d


def sm_config(meta):
    cfg = make_config()
    for k, v in meta["config"].items():
        setattr(cfg, k, v)
    return cfg


def metrics_of(cy, i):
    return {k: float(v) for k, v in zip(cy["meta"]["metric_keys"], cy["metrics"][i]) if not np.isnan(v)}


def set_sm(sm, cy, i):
    sm.current_state = BY_ID[int(cy["sm_before"][i, 0])]
    sm.consecutive_failures = int(cy["sm_before"][i, 1])
    sm._last_clearance, sm._last_clearance_ahead = float(cy["sm_clr"][i, 0]), float(cy["sm_clr"][i, 1])


def test_state_machine_replays_reference_cycles(cycles):
    """feed the recorded outcome of every plan() attempt: the mirror must walk the same states."""
    cy = cycles
    sm = FailSafeStateMachine(sm_config(cy["meta"]))
    for i in range(len(cy["cost"])):
        set_sm(sm, cy, i)
        m, v = metrics_of(cy, i), float(cy["ego"][i, 3])
        n_plan, found = int(cy["n_plan"][i]), bool(cy["found"][i])
        sm._get_planner_config()
        out = sm.update(n_plan == 1 and found, m, v)
        for a in range(1, n_plan):                    # retries: each entered because the state escalated
            if a < n_plan - 1 or not found:
                out = sm.update(False, m, v)
        assert (BY_ID.index(sm.current_state), sm.consecutive_failures) == tuple(cy["sm_after"][i]), f"cycle {i}"


@pytest.mark.gpu
def test_speculative_cycle_replays_reference_closed_loop(cycles):
    from integrated_path_planning_amd.cubic_spline import CubicSpline2D
    from integrated_path_planning_amd.data_structures import EgoVehicleState
    from integrated_path_planning_amd.planner import FrenetPlanner
    from integrated_path_planning_amd.state_machine import SpeculativePlanningCycle

    cy = cycles
    meta = cy["meta"]
    c = meta["config"]
    planner = FrenetPlanner(CubicSpline2D(meta["waypoints_x"], meta["waypoints_y"]), max_speed=c["ego_max_speed"],
                            max_accel=c["ego_max_accel"], max_curvature=c["ego_max_curvature"], dt=c["dt"],
                            d_road_w=c["d_road_w"], max_road_width=c["max_road_width"], robot_radius=meta["ego_radius"],
                            obstacle_radius=c["obstacle_radius"], max_lat_accel=c.get("ego_max_lat_accel", 3.0),
                            k_j=1.0, k_t=1.0, k_d=1.0, k_s_dot=1.0, k_lat=1.0, k_lon=1.0)
    sm = FailSafeStateMachine(sm_config(meta))
    cyc = SpeculativePlanningCycle(planner, sm, ego_target_speed=c["ego_target_speed"])
    names = ["max_speed_error", "max_accel_error", "max_curvature_error", "max_lat_accel_error", "road_bound_error",
             "collision_error", "ok", "stop_distance_error"]
    total_launches = 0
    for i in range(len(cy["cost"])):
        set_sm(sm, cy, i)
        planner._last_kappa = float(cy["last_kappa"][i])
        if np.isnan(cy["prev_s"][i]):
            if hasattr(planner.converter, "_prev_s"):
                del planner.converter._prev_s
        else:
            planner.converter._prev_s = float(cy["prev_s"][i])
        p, t = cy["dyn_shape"][i]
        dyn = cy["dyn"][i, :p, :t]
        ego = EgoVehicleState(*[float(v) for v in cy["ego"][i]])
        r = cyc.execute(ego, np.empty((0, 2)), dyn, metrics_of(cy, i))
        total_launches += 1
        label = f"cycle {i}"
        assert r.attempts == int(cy["n_plan"][i]), label
        assert (r.planned_path is not None) == bool(cy["found"][i]), label
        assert (BY_ID.index(sm.current_state), sm.consecutive_failures) == tuple(cy["sm_after"][i]), label
        np.testing.assert_allclose(planner.converter._prev_s, cy["after_prev_s"][i], atol=1e-9, err_msg=label)
        np.testing.assert_allclose(planner._last_kappa, cy["after_kappa"][i], rtol=1e-8, atol=1e-8, err_msg=label)
        st = cy["stats"][i]
        want_stats = None if st[0] == -2 else {names[k]: int(st[k]) for k in range(8) if st[k] >= 0}
        assert planner.last_check_stats == want_stats, label
        if r.planned_path is not None:
            np.testing.assert_allclose(r.planned_path.cost, cy["cost"][i], rtol=1e-8, err_msg=label)
            head = [r.planned_path.x[1], r.planned_path.y[1], r.planned_path.v[1], r.planned_path.a[1],
                    r.planned_path.c[1]]
            np.testing.assert_allclose(head, cy["head"][i], rtol=1e-8, atol=1e-8, err_msg=label)
    assert total_launches == 274 and int(cy["n_plan"].sum()) == 278      # 278 reference plan() calls in 274 launches

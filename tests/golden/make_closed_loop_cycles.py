#!/usr/bin/env python3
"""Record every _execute_planning_cycle() of a REFERENCE closed-loop run (build container only) -- SURVEY 8(f2).

Same run as make_closed_loop.py (scenario_01, method cv, scripted constant-velocity pedestrians).  Per step:
inputs of the cycle (ego, obstacle tensor, safety metrics, state-machine state before) and its outcome (adopted
path summary, number of plan() calls, state-machine state after, planner state after).  Data only.
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
STATES = {"NORMAL": 0, "CAUTION": 1, "EMERGENCY": 2}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ref", default="/root/reference")
    args = ap.parse_args()
    lg = types.ModuleType("loguru")

    class _Logger:
        def __getattr__(self, name):
            return lambda *a, **k: None

    lg.logger = _Logger()
    sys.modules["loguru"] = lg
    sys.modules["pysocialforce"] = types.ModuleType("pysocialforce")
    sys.path.insert(0, args.ref)
    os.chdir(args.ref)
    from src.config import SimulationConfig
    import src.simulation.integrated_simulator as simmod
    from src.simulation.replay_source import ReplayPedestrianSource

    cfg = yaml.safe_load(open(os.path.join(args.ref, "scenarios", "scenario_01.yaml")))
    peds = np.array(cfg["ped_initial_states"], dtype=float)
    cfg.update(ped_initial_states=[], ped_groups=[], sgan_model_path=None, prediction_method="cv",
               visualization_enabled=False)
    config = SimulationConfig(**cfg)
    sim = simmod.IntegratedSimulator(config)
    n_frames = int(config.total_time / config.dt) + 64
    t = np.arange(n_frames) * config.dt
    sim.pedestrian_sim = ReplayPedestrianSource(peds[None, :, 0:2] + peds[None, :, 2:4] * t[:, None, None], dt=config.dt)
    sim.warmup()

    last_metrics = {}
    orig_metrics = simmod.compute_safety_metrics_static

    def rec_metrics(*a, **k):
        m = orig_metrics(*a, **k)
        last_metrics.clear()
        last_metrics.update(m)
        return m

    simmod.compute_safety_metrics_static = rec_metrics
    n_calls = [0]
    orig_plan = sim.planner.plan

    def counting_plan(*a, **k):
        n_calls[0] += 1
        return orig_plan(*a, **k)

    sim.planner.plan = counting_plan
    steps = []
    orig_cycle = sim._execute_planning_cycle

    def rec_cycle(static_obstacles, dynamic_obstacles, ped_state, dynamic_obstacles_distribution=None):
        sm = sim.state_machine
        e = sim.ego_state
        before = dict(ego=[e.x, e.y, e.yaw, e.v, e.a], sm_state=STATES[sm.current_state.name],
                      sm_fail=sm.consecutive_failures, sm_clr=sm._last_clearance, sm_clr_ahead=sm._last_clearance_ahead,
                      last_kappa=float(sim.planner._last_kappa),
                      prev_s=float(getattr(sim.planner.converter, "_prev_s", np.nan)),
                      dyn=np.array(dynamic_obstacles, dtype=float))
        n_calls[0] = 0
        path, t_plan = orig_cycle(static_obstacles, dynamic_obstacles, ped_state, dynamic_obstacles_distribution)
        assert dynamic_obstacles_distribution is None and len(static_obstacles) == 0
        before.update(metrics={k: (float(v) if not isinstance(v, bool) else bool(v)) for k, v in last_metrics.items()
                               if isinstance(v, (int, float, bool, np.floating, np.bool_))},
                      n_plan=n_calls[0], found=path is not None,
                      cost=float(path.cost) if path is not None else np.inf,
                      head=[path.x[1], path.y[1], path.v[1], path.a[1], path.c[1]] if path is not None else [np.nan] * 5,
                      after_state=STATES[sm.current_state.name], after_fail=sm.consecutive_failures,
                      after_kappa=float(sim.planner._last_kappa), after_prev_s=float(sim.planner.converter._prev_s),
                      stats=sim.planner.last_check_stats)
        steps.append(before)
        return path, t_plan

    sim._execute_planning_cycle = rec_cycle
    sim.run()
    n = len(steps)
    P = max(s["dyn"].shape[0] for s in steps)
    T = max(s["dyn"].shape[1] for s in steps)
    dyn = np.full((n, P, T, 2), np.nan)
    shp = np.zeros((n, 2), np.int32)
    for i, s in enumerate(steps):
        p, tt = s["dyn"].shape[:2]
        dyn[i, :p, :tt] = s["dyn"]
        shp[i] = (p, tt)
    mkeys = sorted({k for s in steps for k in s["metrics"]})
    names = ["max_speed_error", "max_accel_error", "max_curvature_error", "max_lat_accel_error", "road_bound_error",
             "collision_error", "ok", "stop_distance_error"]
    out = dict(
        ego=np.array([s["ego"] for s in steps]), dyn=dyn, dyn_shape=shp,
        sm_before=np.array([[s["sm_state"], s["sm_fail"]] for s in steps], dtype=np.int32),
        sm_clr=np.array([[s["sm_clr"], s["sm_clr_ahead"]] for s in steps]),
        last_kappa=np.array([s["last_kappa"] for s in steps]), prev_s=np.array([s["prev_s"] for s in steps]),
        metrics=np.array([[float(s["metrics"].get(k, np.nan)) for k in mkeys] for s in steps]),
        n_plan=np.array([s["n_plan"] for s in steps], dtype=np.int32), found=np.array([s["found"] for s in steps]),
        cost=np.array([s["cost"] for s in steps]), head=np.array([s["head"] for s in steps]),
        sm_after=np.array([[s["after_state"], s["after_fail"]] for s in steps], dtype=np.int32),
        after_kappa=np.array([s["after_kappa"] for s in steps]), after_prev_s=np.array([s["after_prev_s"] for s in steps]),
        stats=np.array([[-2] * 8 if s["stats"] is None else [s["stats"].get(k, -1) for k in names] for s in steps],
                       dtype=np.int32),
        meta=np.array(json.dumps(dict(metric_keys=mkeys, config={k: v for k, v in cfg.items()
                                                                  if isinstance(v, (int, float, str, bool)) or v is None},
                                      waypoints_x=list(config.reference_waypoints_x),
                                      waypoints_y=list(config.reference_waypoints_y),
                                      ego_radius=float(sim.ego_radius)))))
    path = os.path.join(HERE, "closed_loop", "scenario01_cv_cycles.npz")
    np.savez_compressed(path, **out)
    print(f"{n} cycles, plan() calls {int(out['n_plan'].sum())}, retries {int((out['n_plan'] - 1).sum())}, "
          f"found {int(out['found'].sum())}, states after: {np.bincount(out['sm_after'][:, 0]).tolist()}, "
          f"{os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()

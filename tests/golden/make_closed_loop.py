#!/usr/bin/env python3
"""Record every FrenetPlanner.plan() call of a REFERENCE closed-loop run (build container only).

BASELINE config 1: scenarios/scenario_01.yaml, prediction method 'cv', one ego.  pysocialforce is
not installable offline, so the crowd is the SURVEY 8(c) substitute: the scenario's 14 pedestrians
move with their initial constant velocities (ReplayPedestrianSource).  The reference simulator,
state machine, predictor and planner are imported read-only; `loguru` and `pysocialforce` are
replaced by empty in-process modules.  Output: tests/golden/closed_loop/scenario01_cv.npz holding,
per plan() call, the inputs (ego, planner state, target speed, overrides, stop directive,
obstacle tensors) and the outputs (found, cost, best path head, last_check_stats) -- data only.
"""
import argparse
import json
import os
import sys
import time
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
STATUS_NAMES = ["max_speed_error", "max_accel_error", "max_curvature_error", "max_lat_accel_error",
                "road_bound_error", "collision_error", "ok", "stop_distance_error"]


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
    from src.simulation.integrated_simulator import IntegratedSimulator
    from src.simulation.replay_source import ReplayPedestrianSource

    cfg = yaml.safe_load(open(os.path.join(args.ref, "scenarios", "scenario_01.yaml")))
    peds = np.array(cfg["ped_initial_states"], dtype=float)
    cfg["ped_initial_states"] = []
    cfg["ped_groups"] = []
    cfg["sgan_model_path"] = None
    cfg["prediction_method"] = "cv"
    cfg["visualization_enabled"] = False
    config = SimulationConfig(**cfg)
    sim = IntegratedSimulator(config)
    n_frames = int(config.total_time / config.dt) + 64
    t = np.arange(n_frames) * config.dt
    traj = peds[None, :, 0:2] + peds[None, :, 2:4] * t[:, None, None]
    sim.pedestrian_sim = ReplayPedestrianSource(traj, dt=config.dt)
    sim.warmup()

    planner = sim.planner
    calls = []
    orig_plan = planner.plan

    def recording_plan(ego_state, static_obstacles, dynamic_obstacles=None, target_speed=30.0 / 3.6,
                       constraint_overrides=None, dynamic_obstacles_distribution=None, max_stop_distance=None):
        rec = dict(ego=[ego_state.x, ego_state.y, ego_state.yaw, ego_state.v, ego_state.a],
                   last_kappa=float(planner._last_kappa),
                   prev_s=float(getattr(planner.converter, "_prev_s", np.nan)),
                   target_speed=float(target_speed), overrides=constraint_overrides,
                   max_stop=None if max_stop_distance is None else float(max_stop_distance),
                   static=np.array(static_obstacles, dtype=float).reshape(-1, 2),
                   dyn=None if dynamic_obstacles is None else np.array(dynamic_obstacles, dtype=float))
        assert dynamic_obstacles_distribution is None
        t0 = time.perf_counter()
        path = orig_plan(ego_state, static_obstacles, dynamic_obstacles, target_speed=target_speed,
                         constraint_overrides=constraint_overrides,
                         dynamic_obstacles_distribution=dynamic_obstacles_distribution,
                         max_stop_distance=max_stop_distance)
        rec["ms"] = (time.perf_counter() - t0) * 1e3
        rec["found"] = path is not None
        rec["cost"] = float(path.cost) if path is not None else np.inf
        rec["n_keep"] = len(path.x) if path is not None else 0
        rec["head"] = ([path.x[1], path.y[1], path.yaw[1], path.v[1], path.a[1], path.c[1]]
                       if path is not None and len(path.x) > 1 else [np.nan] * 6)
        rec["tail"] = ([path.x[-1], path.y[-1], path.s[-1], path.d[-1], path.v[-1]]
                       if path is not None else [np.nan] * 5)
        st = planner.last_check_stats
        rec["stats"] = [-2] * 8 if st is None else [st.get(k, -1) for k in STATUS_NAMES]
        rec["last_kappa_after"] = float(planner._last_kappa)
        rec["prev_s_after"] = float(planner.converter._prev_s)
        calls.append(rec)
        return path

    planner.plan = recording_plan
    sim.run()
    n = len(calls)
    P = max(c["dyn"].shape[0] for c in calls if c["dyn"] is not None)
    T = max(c["dyn"].shape[1] for c in calls if c["dyn"] is not None)
    dyn = np.full((n, P, T, 2), np.nan)
    dyn_shape = np.zeros((n, 2), dtype=np.int32)
    for i, c in enumerate(calls):
        if c["dyn"] is not None and c["dyn"].size:
            p, tt = c["dyn"].shape[:2]
            dyn[i, :p, :tt] = c["dyn"]
            dyn_shape[i] = (p, tt)
        assert c["static"].shape[0] == 0
    ov_keys = ["max_speed", "max_accel", "max_curvature", "max_lat_accel"]
    out = dict(
        ego=np.array([c["ego"] for c in calls]), last_kappa=np.array([c["last_kappa"] for c in calls]),
        prev_s=np.array([c["prev_s"] for c in calls]), target_speed=np.array([c["target_speed"] for c in calls]),
        overrides=np.array([[np.nan if not c["overrides"] else c["overrides"].get(k, np.nan) for k in ov_keys]
                            for c in calls]),
        max_stop=np.array([np.nan if c["max_stop"] is None else c["max_stop"] for c in calls]),
        dyn=dyn.astype(np.float64), dyn_shape=dyn_shape,
        found=np.array([c["found"] for c in calls]), cost=np.array([c["cost"] for c in calls]),
        n_keep=np.array([c["n_keep"] for c in calls], dtype=np.int32), head=np.array([c["head"] for c in calls]),
        tail=np.array([c["tail"] for c in calls]), stats=np.array([c["stats"] for c in calls], dtype=np.int32),
        last_kappa_after=np.array([c["last_kappa_after"] for c in calls]),
        prev_s_after=np.array([c["prev_s_after"] for c in calls]), ref_ms=np.array([c["ms"] for c in calls]),
        meta=np.array(json.dumps(dict(
            planner=dict(max_speed=config.ego_max_speed, max_accel=config.ego_max_accel,
                         max_curvature=config.ego_max_curvature, max_lat_accel=getattr(config, "ego_max_lat_accel", 3.0),
                         dt=config.dt, d_road_w=config.d_road_w, max_road_width=config.max_road_width,
                         robot_radius=sim.ego_radius, obstacle_radius=config.obstacle_radius,
                         min_t=getattr(config, "min_t", 4.0), max_t=getattr(config, "max_t", 5.0),
                         d_t_s=getattr(config, "d_t_s", 5.0 / 3.6), k_j=config.k_j, k_t=config.k_t, k_d=config.k_d,
                         k_s_dot=config.k_s_dot, k_lat=config.k_lat, k_lon=config.k_lon,
                         chance_epsilon=getattr(config, "chance_epsilon", 0.0),
                         collision_margin_inflation=getattr(config, "collision_margin_inflation", 1.0)),
            footprint=None if sim.ego_footprint is None else dict(offsets=[float(v) for v in sim.ego_footprint.offsets],
                                                                   radius=float(sim.ego_footprint.radius)),
            waypoints_x=list(config.reference_waypoints_x), waypoints_y=list(config.reference_waypoints_y),
            steps=len(sim.history) if hasattr(sim, "history") else None,
            termination=sim.termination_reason))))
    os.makedirs(os.path.join(HERE, "closed_loop"), exist_ok=True)
    path = os.path.join(HERE, "closed_loop", "scenario01_cv.npz")
    np.savez_compressed(path, **out)
    ms = out["ref_ms"]
    print(f"{n} plan() calls, found {int(out['found'].sum())}, termination {sim.termination_reason}; "
          f"reference plan ms: mean {ms.mean():.1f} p50 {np.percentile(ms, 50):.1f} p95 {np.percentile(ms, 95):.1f}; "
          f"{os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()

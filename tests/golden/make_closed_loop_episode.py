#!/usr/bin/env python3
"""Whole closed-loop episodes of the REFERENCE simulator (build container only) -- SURVEY 8(f4).

IntegratedSimulator.run() on scenario_01 (method cv) with its pedestrians replayed as scripted constant-velocity
tracks through the reference's own ReplayPedestrianSource (pysocialforce is not installable offline: the SURVEY 8(c)
substitute), in three variants of the pedestrian script.  Recorded per step: what save_results() writes to
trajectory.npz (integrated_simulator.py:906-982) plus acceleration, safety metrics and state.  Data only.
"""
import argparse
import json
import os
import sys
import tempfile
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
STATES = {"NORMAL": 0, "CAUTION": 1, "EMERGENCY": 2}
VARIANTS = {"base": dict(scenario="scenario_01", speed=1.0, dy=0.0), "fast": dict(scenario="scenario_01", speed=1.3, dy=0.0),
            "shift": dict(scenario="scenario_01", speed=1.0, dy=1.0),
            "walls": dict(scenario="scenario_02", speed=1.0, dy=0.0),        # static obstacle rectangles either side
            "turn": dict(scenario="scenario_03", speed=1.0, dy=0.0),         # curved reference path
            # the multi-circle ego footprint (planner collision geometry AND safety metrics, footprint.py) in the loop
            "footprint": dict(scenario="scenario_01", speed=1.0, dy=0.5,
                              cfg=dict(ego_footprint="multi_circle", ego_footprint_n_circles=3)),
            # the planner's single-sample dynamic margin inflated (frenet_planner.py:1126-1179), metrics unaffected
            "inflate": dict(scenario="scenario_01", speed=1.15, dy=-0.5, cfg=dict(collision_margin_inflation=1.2))}
# random pedestrian scripts: every pedestrian's start jittered by N(0, 0.7 m), its velocity scaled by U(0.7, 1.4) and
# turned by N(0, 0.15 rad), the whole crowd shifted -- seeds fixed here so that the fixture can be regenerated
for _k, (_sc, _seed) in enumerate([("scenario_01", 11), ("scenario_01", 12), ("scenario_03", 13), ("scenario_02", 14),
                                   ("scenario_03", 15), ("scenario_01", 16)]):
    VARIANTS[f"rnd{_k}"] = dict(scenario=_sc, speed=1.0, dy=0.0, jitter_seed=_seed)


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

    out = {}
    meta = {"variants": {}, "states": STATES}
    for name, var in VARIANTS.items():
        raw = yaml.safe_load(open(os.path.join(args.ref, "scenarios", var["scenario"] + ".yaml")))
        peds0 = np.array(raw["ped_initial_states"], dtype=float)
        cfg = dict(raw)
        cfg.update(ped_initial_states=[], ped_groups=[], sgan_model_path=None, prediction_method="cv",
                   visualization_enabled=False)
        cfg.update(var.get("cfg", {}))
        config = SimulationConfig(**cfg)
        sim = simmod.IntegratedSimulator(config)
        peds = peds0.copy()
        peds[:, 2:4] *= var["speed"]
        peds[:, 1] += var["dy"]
        if "jitter_seed" in var and len(peds):
            rng = np.random.default_rng(var["jitter_seed"])
            peds[:, 0:2] += rng.normal(0.0, 0.7, (len(peds), 2)) + rng.uniform(-1.5, 1.5, 2)
            ang = rng.normal(0.0, 0.15, len(peds))
            vx, vy = peds[:, 2].copy(), peds[:, 3].copy()
            sc = rng.uniform(0.7, 1.4, len(peds))
            peds[:, 2] = sc * (np.cos(ang) * vx - np.sin(ang) * vy)
            peds[:, 3] = sc * (np.sin(ang) * vx + np.cos(ang) * vy)
        n_frames = int(config.total_time / config.dt) + 64
        t = np.arange(n_frames) * config.dt
        traj = peds[None, :, 0:2] + peds[None, :, 2:4] * t[:, None, None]
        sim.pedestrian_sim = ReplayPedestrianSource(traj, dt=config.dt)
        sim.warmup()
        sim.run()
        h = sim.history
        n = len(h)
        with tempfile.TemporaryDirectory() as td:
            sim.visualize = lambda *a, **k: None
            try:
                sim.save_results(td)
            except Exception as e:                       # plotting / metrics extras are not part of the fixture
                print("save_results:", type(e).__name__, e)
            z = np.load(os.path.join(td, "trajectory.npz"), allow_pickle=True)   # written by this very run
            keys = {k: [str(z[k].dtype), list(z[k].shape)] for k in z.files}
        L = 64
        px = np.full((n, L), np.nan); py = np.full((n, L), np.nan)
        plen = np.zeros(n, np.int32)
        for i, r in enumerate(h):
            if r.planned_path is not None:
                m = len(r.planned_path.x)
                plen[i] = m
                px[i, :m] = r.planned_path.x; py[i, :m] = r.planned_path.y
        pre = f"{name}_"
        out[pre + "ped_traj"] = traj
        out[pre + "times"] = np.array([r.time for r in h])
        out[pre + "ego"] = np.array([[r.ego_state.x, r.ego_state.y, r.ego_state.yaw, r.ego_state.v, r.ego_state.a,
                                      r.ego_state.jerk] for r in h])
        out[pre + "state"] = np.array([STATES[r.ego_state.state.name] for r in h], dtype=np.int32)
        out[pre + "metrics"] = np.array([[r.metrics.get("min_distance", np.inf), r.metrics.get("ttc", np.inf),
                                          r.metrics.get("clearance", np.inf), r.metrics.get("clearance_ahead", np.inf),
                                          float(r.metrics.get("collision", False)),
                                          r.metrics.get("n_collision_rejected", -1)] for r in h])
        out[pre + "planned_cost"] = np.array([r.planned_path.cost if r.planned_path is not None else np.inf for r in h])
        out[pre + "planned_len"] = plen
        out[pre + "planned_x"] = px
        out[pre + "planned_y"] = py
        out[pre + "pred_shape"] = np.array([list(r.predicted_trajectories.shape) if r.predicted_trajectories is not None
                                            else [0, 0, 0] for r in h], dtype=np.int32)
        out[pre + "pred_first"] = np.array([r.predicted_trajectories[0, :3].ravel() if r.predicted_trajectories is not None
                                            else np.full(6, np.nan) for r in h])
        # the RESOLVED configuration (scenario values + the defaults of the reference's SimulationConfig)
        resolved = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in vars(config).items()}
        resolved = {k: v for k, v in resolved.items() if isinstance(v, (int, float, str, bool, list)) or v is None}
        meta["variants"][name] = dict(steps=n, termination=sim.termination_reason, npz_keys=keys, config=resolved,
                                      ego_radius=float(sim.ego_radius), ped_radius=float(sim.ped_radius),
                                      n_static_points=int(len(sim.static_obstacle_points)),
                                      **{k: v for k, v in var.items() if k != "cfg"})
        print(name, n, "steps,", sim.termination_reason, "states", np.bincount(out[pre + "state"], minlength=3).tolist(),
              "no path", int((plen == 0).sum()))
    meta["config"] = meta["variants"]["base"]["config"]          # scenario_01, kept for the callers that read it here
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, "closed_loop", "reference_cv_episodes.npz")
    np.savez_compressed(path, **out)
    print(f"{os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Distribution-aware closed-loop episodes of the REFERENCE simulator (build container only) -- SURVEY 8(f4).

IntegratedSimulator.run() on scenario_01 with `distribution_aware_planning`: every step the planner receives the whole
sampled prediction distribution and applies its chance constraint (integrated_simulator.py:459-460, 514-525, 576-584;
frenet_planner.py:1076-1124).  The reference's only multi-sample source is Social-GAN, whose weights are not available
offline; its ONE forward pass (TrajectoryPredictor.predict) is therefore replaced by a scripted sample generator
(tests/closed_loop_common.py::scripted_raw_sample) whose output still runs through the reference's own
process_prediction, predict_single_best, prepend logic, planner and fail-safe loop.  Pedestrians are replayed scripted
tracks through the reference's ReplayPedestrianSource, as in make_closed_loop_episode.py.  Data only.
"""
import argparse
import json
import os
import sys
import types

import numpy as np
import yaml

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))                      # tests/: closed_loop_common
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))     # repo root
STATES = {"NORMAL": 0, "CAUTION": 1, "EMERGENCY": 2}
VARIANTS = {"s6_eps02": dict(n_samples=6, chance_epsilon=0.2, speed=1.0, dy=0.0),      # floor(0.2 * 6) = 1 sample may collide
            "s4_eps0": dict(n_samples=4, chance_epsilon=0.0, speed=1.15, dy=0.5),      # robust: no sample may collide
            "s5_best_only": dict(n_samples=5, chance_epsilon=0.0, speed=1.0, dy=0.0, aware=False)}   # plans on the best sample


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
    from closed_loop_common import scripted_raw_sample
    sys.path.insert(0, args.ref)
    os.chdir(args.ref)
    from src.config import SimulationConfig
    import src.simulation.integrated_simulator as simmod
    from src.simulation.replay_source import ReplayPedestrianSource

    out, meta = {}, {"variants": {}, "states": STATES}
    for name, var in VARIANTS.items():
        raw = yaml.safe_load(open(os.path.join(args.ref, "scenarios", "scenario_01.yaml")))
        peds0 = np.array(raw["ped_initial_states"], dtype=float)
        cfg = dict(raw)
        cfg.update(ped_initial_states=[], ped_groups=[], sgan_model_path=None, prediction_method="cv",
                   visualization_enabled=False, chance_epsilon=var["chance_epsilon"])
        config = SimulationConfig(**cfg)
        sim = simmod.IntegratedSimulator(config)
        S = var["n_samples"]
        aware = var.get("aware", True)
        sim.distribution_aware_planning = aware
        pr = sim.predictor
        pr.num_samples = S
        calls = {"k": 0}

        def scripted_predict(obs_traj, obs_traj_rel, seq_start_end, staleness=0.0, pr=pr, calls=calls, S=S):
            k = calls["k"] % S
            calls["k"] += 1
            obs = obs_traj.cpu().numpy().astype(np.float64)      # the observer's float32 tensors, widened
            raw_k = scripted_raw_sample(obs[-1], obs[-2], k, S, pr.pred_len, pr.sgan_dt)
            return pr.process_prediction(raw_k, anchor_pos=obs[-1], staleness=staleness)

        pr.predict = scripted_predict
        peds = peds0.copy()
        peds[:, 2:4] *= var["speed"]
        peds[:, 1] += var["dy"]
        n_frames = int(config.total_time / config.dt) + 64
        t = np.arange(n_frames) * config.dt
        traj = peds[None, :, 0:2] + peds[None, :, 2:4] * t[:, None, None]
        sim.pedestrian_sim = ReplayPedestrianSource(traj, dt=config.dt)
        sim.warmup()
        sim.run()
        h = sim.history
        n = len(h)
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
        resolved = {k: (v.tolist() if isinstance(v, np.ndarray) else v) for k, v in vars(config).items()}
        resolved = {k: v for k, v in resolved.items() if isinstance(v, (int, float, str, bool, list)) or v is None}
        resolved["distribution_aware_planning"] = aware
        resolved["num_samples"] = S
        meta["variants"][name] = dict(steps=n, termination=sim.termination_reason, config=resolved,
                                      ego_radius=float(sim.ego_radius), ped_radius=float(sim.ped_radius),
                                      n_static_points=int(len(sim.static_obstacle_points)), predict_calls=calls["k"], **var)
        print(name, n, "steps,", sim.termination_reason, "states", np.bincount(out[pre + "state"], minlength=3).tolist(),
              "no path", int((plen == 0).sum()), "predict calls", calls["k"])
    out["meta"] = np.array(json.dumps(meta))
    path = os.path.join(HERE, "closed_loop", "reference_dist_episodes.npz")
    np.savez_compressed(path, **out)
    print(f"{os.path.getsize(path) / 1e6:.2f} MB")


if __name__ == "__main__":
    main()

"""Batched closed-loop driver: many simulator episodes in lock-step on one GPU.

One episode is what ``IntegratedSimulator.step()/run()`` (src/simulation/integrated_simulator.py:678-892) does for one
ego vehicle: advance the pedestrians, sample the observer, predict, prepend the current positions, safety metrics,
the escalate-and-retry planning cycle, ego update or emergency stop, termination on collision / goal / timeout.  Here
N episodes that share planner parameters and reference path advance together, and every step issues

* ONE constant-velocity prediction launch over the pedestrians of all running episodes (row f1),
* ONE safety-metrics launch before planning and one after the ego update (row f3),
* ONE ``fot_plan_batch`` holding every escalation level of every episode (row f2),
* ONE nearest-point launch for the goal test,

instead of N x (1 + up to 3 retries) sequential ``plan()`` calls.  Pedestrians are replayed tracks -- the contract of the
reference's ``ReplayPedestrianSource`` (src/simulation/replay_source.py:31-118); the Social-Force simulator and the
Social-GAN network are outside SURVEY 8.  ``save_results`` writes ``trajectory.npz`` with the reference's keys, dtypes
and array shapes (integrated_simulator.py:906-982), so the existing analysis scripts read it unchanged.

The control logic on the host is the reference's scalar logic; all arithmetic on candidate paths, predictions and
metrics runs in libfot.
"""
from __future__ import annotations

import copy
import os
import time
from collections import deque
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from .batch import PlanRequest
from .data_structures import EgoVehicleState, FrenetPath
from .footprint import EgoFootprint
from .planner import BatchPlanner
from .prediction import PredictionResampler
from .state_machine import FailSafeStateMachine, SpeculativePlanningCycle, VehicleState


class ReplayPedestrians:
    """Frame-by-frame replay of [T, N, 2] tracks: step()/get_state() of replay_source.py:31-118."""

    def __init__(self, trajectories, dt: float, velocities=None, goals=None):
        traj = np.asarray(trajectories, dtype=float)
        if traj.ndim != 3 or traj.shape[2] != 2:
            raise ValueError(f"trajectories must be [T, N, 2], got shape {traj.shape}")
        self.trajectories = traj
        self.n_frames, self.n_peds, _ = traj.shape
        self.dt = float(dt)
        self.time = 0.0
        self._idx = 0
        if velocities is not None:
            self.velocities = np.asarray(velocities, dtype=float)
        else:                                            # forward difference, last step repeats (:77-84)
            vel = np.zeros_like(traj)
            if traj.shape[0] >= 2:
                vel[:-1] = (traj[1:] - traj[:-1]) / self.dt
                vel[-1] = vel[-2]
            self.velocities = vel
        self.goals = np.asarray(goals, dtype=float) if goals is not None else traj[-1].copy()

    def step(self, n: int = 1) -> None:
        for _ in range(n):
            if self._idx < self.n_frames - 1:
                self._idx += 1
            self.time += self.dt

    @property
    def positions(self) -> np.ndarray:
        return self.trajectories[self._idx]

    @property
    def current_velocities(self) -> np.ndarray:
        return self.velocities[self._idx]


class Observer:
    """Sliding window sampled every sgan_dt of pedestrian time (src/pedestrian/observer.py:28-102)."""

    def __init__(self, obs_len: int, dt: float, sgan_dt: float = 0.4):
        self.obs_len, self.dt, self.sgan_dt = obs_len, dt, sgan_dt
        self.history: deque = deque(maxlen=obs_len)
        self.timestamps: deque = deque(maxlen=obs_len)
        self.accumulated_time = 0.0
        self._last_update_timestamp: Optional[float] = None

    def update(self, positions: np.ndarray, timestamp: float) -> None:
        delta_t = self.dt if self._last_update_timestamp is None else max(timestamp - self._last_update_timestamp, 0.0)
        self._last_update_timestamp = timestamp
        self.accumulated_time += delta_t
        if self.accumulated_time + 1e-9 >= self.sgan_dt:
            self.history.append(positions.copy())
            self.timestamps.append(timestamp)
            self.accumulated_time = max(self.accumulated_time - self.sgan_dt, 0.0)

    @property
    def is_ready(self) -> bool:
        return len(self.history) >= self.obs_len

    @property
    def last_sample_time(self) -> Optional[float]:
        return self.timestamps[-1] if self.timestamps else None


class _PlannerState:
    """The per-episode state a FrenetPlanner keeps between calls; the engine is shared by all episodes."""

    class _Conv:
        _prev_s: Optional[float] = None

    def __init__(self, engine: BatchPlanner):
        self.engine = engine
        self.converter = _PlannerState._Conv()
        self._last_kappa = 0.0
        self.last_check_stats = None

    def reset_ego_curvature(self) -> None:
        self._last_kappa = 0.0


@dataclass
class StepRecord:
    """What SimulationResult holds of one step (data_structures.py:256-281), as plain arrays."""
    time: float
    ego: EgoVehicleState
    ped_positions: np.ndarray
    ped_velocities: np.ndarray
    ped_goals: np.ndarray
    predicted_trajectories: Optional[np.ndarray]
    planned_path: Optional[FrenetPath]
    metrics: Dict[str, Any]
    processing_times: Dict[str, float]


@dataclass
class Episode:
    peds: ReplayPedestrians
    observer: Observer
    ego: EgoVehicleState
    sm: FailSafeStateMachine
    pstate: _PlannerState
    cycle: SpeculativePlanningCycle
    goal_prev_s: Optional[float] = None          # nearest-point cache of the simulator's own converter (:873)
    last_clearance: float = float("inf")
    time: float = 0.0
    step_count: int = 0
    history: List[StepRecord] = field(default_factory=list)
    termination_reason: Optional[str] = None


def _cfg(config, name, default=None):
    return config.get(name, default) if isinstance(config, dict) else getattr(config, name, default)


class _Cfg:
    """getattr view of a dict (the reference's classes read their configuration with getattr)."""

    def __init__(self, d):
        self.__dict__.update(d)


def expand_static_obstacles(static_obstacles, step: float = 0.5) -> np.ndarray:
    """Rectangles [x_min, x_max, y_min, y_max] -> boundary points every `step` (integrated_simulator.py:805-832)."""
    if static_obstacles is None or len(static_obstacles) == 0:
        return np.empty((0, 2))
    points = []
    for rect in static_obstacles:
        if len(rect) != 4:
            continue
        x_min, x_max, y_min, y_max = rect
        xs = np.arange(x_min, x_max + step, step)
        ys = np.arange(y_min, y_max + step, step)
        for x in xs:
            points.append((x, y_min))
            points.append((x, y_max))
        for y in ys:
            points.append((x_min, y))
            points.append((x_max, y))
    if len(points) == 0:
        return np.empty((0, 2))
    return np.unique(np.array(points), axis=0)


def footprint_from_config(config) -> Optional[EgoFootprint]:
    """src/core/footprint.py footprint_from_config: None = legacy single circle."""
    mode = _cfg(config, "ego_footprint", None)
    if mode is None or mode == "circle":
        return None
    return EgoFootprint.multi_circle(_cfg(config, "vehicle_length"), _cfg(config, "vehicle_width"),
                                     int(_cfg(config, "ego_footprint_n_circles")))


class BatchedClosedLoop:
    """N episodes of the reference's closed loop in lock-step.

    config: the scenario dictionary (or an object with the same attributes) the reference's SimulationConfig is
    built from; ped_tracks: one [T, N_i, 2] array of replayed pedestrian positions per episode (frame spacing
    config.dt, frame 0 = time 0 before warm-up); ego_initial_states: optional per-episode [x, y, yaw, v, a].
    """

    def __init__(self, config, ped_tracks: Sequence[np.ndarray], ego_initial_states: Optional[Sequence] = None,
                 device: int = -1, engine=None, resampler=None):
        self.config = config if not isinstance(config, dict) else _Cfg(config)
        c = self.config
        self.dt = float(c.dt)
        self.ego_radius = getattr(c, "ego_radius", 1.0)
        self.ped_radius = getattr(c, "ped_radius", 0.3)
        self.footprint = footprint_from_config(c)
        if getattr(c, "prediction_method", "sgan") != "cv":
            raise NotImplementedError("only the constant-velocity predictor is part of this build (SURVEY 8 f1)")
        if getattr(c, "distribution_aware_planning", False):
            raise NotImplementedError("the cv predictor yields one sample: no distribution to plan against")
        self.static_obstacle_points = expand_static_obstacles(getattr(c, "static_obstacles", None), step=0.5)
        # engine / resampler: objects with BatchPlanner's / PredictionResampler's methods; the tests drive the
        # host logic with stand-ins when there is no GPU, the product always builds the libfot handle below
        self._owns_engine = engine is None
        self.engine = engine if engine is not None else BatchPlanner(
            waypoints=(np.asarray(c.reference_waypoints_x, float), np.asarray(c.reference_waypoints_y, float)),
            device=device, max_speed=c.ego_max_speed, max_accel=c.ego_max_accel, max_curvature=c.ego_max_curvature,
            max_lat_accel=getattr(c, "ego_max_lat_accel", 3.0), dt=c.dt, d_road_w=c.d_road_w,
            max_road_width=c.max_road_width, robot_radius=self.ego_radius, obstacle_radius=c.obstacle_radius,
            min_t=getattr(c, "min_t", 4.0), max_t=getattr(c, "max_t", 5.0), d_t_s=getattr(c, "d_t_s", 5.0 / 3.6),
            n_s_sample=getattr(c, "n_s_sample", 1), k_j=c.k_j, k_t=c.k_t, k_d=c.k_d, k_s_dot=c.k_s_dot, k_lat=c.k_lat,
            k_lon=c.k_lon, chance_epsilon=getattr(c, "chance_epsilon", 0.0),
            collision_margin_inflation=getattr(c, "collision_margin_inflation", 1.0), footprint=self.footprint)
        self.s_end = float(self.engine.path_coeffs()[0][-1])
        self.sgan_dt = 0.4                                            # integrated_simulator.py:323-327
        self.resampler = resampler if resampler is not None else PredictionResampler(
            self.engine, pred_len=c.pred_len, sgan_dt=self.sgan_dt, sim_dt=c.dt, plan_horizon=getattr(c, "max_t", 5.0))
        n = len(ped_tracks)
        if ego_initial_states is None:
            ego_initial_states = [c.ego_initial_state] * n
        self.episodes: List[Episode] = []
        for tracks, e0 in zip(ped_tracks, ego_initial_states):
            e0 = np.asarray(e0, float)
            sm = FailSafeStateMachine(c)
            ps = _PlannerState(self.engine)
            ego = EgoVehicleState(x=e0[0], y=e0[1], yaw=e0[2], v=e0[3], a=e0[4], jerk=e0[5] if len(e0) > 5 else 0.0,
                                  timestamp=0.0)
            ego.state = sm.current_state
            ep = Episode(peds=ReplayPedestrians(tracks, c.dt), observer=Observer(c.obs_len, c.dt, self.sgan_dt),
                         ego=ego, sm=sm, pstate=ps,
                         cycle=SpeculativePlanningCycle(ps, sm, c.ego_target_speed, max_replan_attempts=3))
            if engine is None:
                ep.cycle._path_kw = {"as_arrays": True}
            self.episodes.append(ep)
        self._warmup()

    def close(self) -> None:
        """Release the libfot handle (streams, workspace) now rather than at garbage collection."""
        if self._owns_engine and self.engine is not None:
            self.engine.close()
        self.engine = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------------------------------------------
    def _warmup(self) -> None:
        """integrated_simulator.py:406-422: fill the observers before time 0."""
        c = self.config
        steps = int(c.obs_len * self.sgan_dt / c.dt)
        for ep in self.episodes:
            for _ in range(steps):
                ep.peds.step()
                ep.observer.update(ep.peds.positions, ep.peds.time)

    @property
    def running(self) -> List[Episode]:
        return [ep for ep in self.episodes if ep.termination_reason is None]

    def _metrics(self, eps: List[Episode]) -> List[Dict[str, Any]]:
        egos = [[ep.ego.x, ep.ego.y, ep.ego.yaw, ep.ego.v] for ep in eps]
        m = self.engine.safety_metrics(egos, [ep.peds.positions for ep in eps],
                                       [ep.peds.current_velocities for ep in eps], self.ego_radius, self.ped_radius,
                                       use_footprint=self.footprint is not None)
        return [{"min_distance": float(r["min_distance"]), "collision": bool(r["collision"]), "ttc": float(r["ttc"]),
                 "clearance": float(r["clearance"]), "clearance_ahead": float(r["clearance_ahead"])} for r in m]

    def _predict(self, eps: List[Episode]):
        """_update_prediction (:424-527) for every episode; one CV launch over all ready episodes' pedestrians."""
        preds: List[Optional[np.ndarray]] = [None] * len(eps)
        dyns: List[np.ndarray] = [None] * len(eps)
        ready = [i for i, ep in enumerate(eps) if ep.observer.is_ready]
        t0 = time.perf_counter()
        if ready:
            groups: Dict[float, List[int]] = {}
            for i in ready:                                          # lock-step episodes share their staleness
                ep = eps[i]
                last = ep.observer.last_sample_time
                stale = max(ep.peds.time - last, 0.0) if last is not None else 0.0
                groups.setdefault(stale, []).append(i)
            for stale, idx in groups.items():
                obs = [np.stack(list(eps[i].observer.history)[-2:], axis=0) for i in idx]   # CV reads the last two samples
                cat = np.concatenate(obs, axis=1)                    # [obs_len, sum P, 2]
                out = self.resampler.predict_cv(cat, staleness=stale, float32_observations=True)
                o = 0
                for i, ob in zip(idx, obs):
                    preds[i] = out[o:o + ob.shape[1]]
                    o += ob.shape[1]
        t_pred = (time.perf_counter() - t0) / max(len(ready), 1)
        for i, ep in enumerate(eps):
            cur = ep.peds.positions[:, None, :]
            d = preds[i] if preds[i] is not None else cur              # not ready: current positions only (:495-498)
            # np.allclose(d[:, 0], cur[:, 0]) of the reference (rtol 1e-5, atol 1e-8; finite inputs), without its overhead
            if preds[i] is not None and not (d.shape[1] >= 1 and bool(
                    np.all(np.abs(d[:, 0, :] - cur[:, 0, :]) <= 1e-8 + 1e-5 * np.abs(cur[:, 0, :])))):
                d = np.concatenate([cur, d], axis=1)                    # prepend the t=0 positions (:503-511)
            dyns[i] = d
        return preds, dyns, t_pred

    def _apply_emergency_stop(self, ep: Episode, old_a: float) -> None:
        """integrated_simulator.py:749-802."""
        c = self.config
        ego = copy.copy(ep.ego)
        cap = getattr(c, "ego_emergency_decel", None)
        if cap is None:
            cap = c.ego_max_accel * 2.0
        clearance = ep.last_clearance
        if np.isfinite(clearance):
            stop_room = max(clearance - 0.2, 0.05)
            required = ego.v ** 2 / (2.0 * stop_room)
        else:
            required = cap
        max_dec = float(np.clip(required, c.ego_max_accel, cap))
        ego.x += ego.v * np.cos(ego.yaw) * c.dt
        ego.y += ego.v * np.sin(ego.yaw) * c.dt
        ego.v = max(0.0, ego.v - max_dec * c.dt)
        new_a = -max_dec if ego.v > 0 else 0.0
        ego.jerk = (new_a - old_a) / c.dt
        ego.a = new_a
        ego.timestamp = ep.time + c.dt
        ep.ego = ego
        ep.pstate.reset_ego_curvature()

    def _update_ego(self, ep: Episode, path: Optional[FrenetPath]) -> None:
        """integrated_simulator.py:655-676."""
        old_a = ep.ego.a
        if path is not None and len(path) >= 2:
            ego = path.get_state_at_index(1)
            ego.jerk = (ego.a - old_a) / self.dt
            ego.timestamp = ep.time + self.dt
            ego.state = ep.sm.current_state
            ep.ego = ego
        else:
            self._apply_emergency_stop(ep, old_a)
            ep.ego.state = ep.sm.current_state

    # ------------------------------------------------------------------------------------------------------
    def step(self) -> int:
        """One lock step of every running episode (integrated_simulator.py:678-747); returns how many ran."""
        eps = self.running
        if not eps:
            return 0
        for ep in eps:                                                # 1. pedestrians + observer
            ep.peds.step()
            ep.observer.update(ep.peds.positions, ep.peds.time)
        preds, dyns, t_pred = self._predict(eps)                      # 2. prediction
        metrics = self._metrics(eps)                                  # 3. planning cycle (:529-653)
        t0 = time.perf_counter()
        reqs: List[PlanRequest] = []
        plans = []
        static = self.static_obstacle_points                          # (fot_batch.static_off is a prefix array: one copy per request)
        for ep, dyn, m in zip(eps, dyns, metrics):
            ep.last_clearance = m.get("clearance_ahead", m.get("clearance", float("inf")))
            ladder, r, budget = ep.cycle.prepare(ep.ego, static, dyn, m)
            plans.append((len(reqs), ladder, budget))
            reqs.extend(r)
        res = self.engine.plan_batch(reqs)
        t_plan = (time.perf_counter() - t0) / len(eps)
        for ep, (base, ladder, budget), m, pred in zip(eps, plans, metrics, preds):
            ego_before = ep.ego
            out = ep.cycle.finish(ladder, budget, res, base, ego_before, m)
            if out.retries:                                           # the retries re-label the current state (:613-615)
                ep.ego = copy.copy(ep.ego)
                ep.ego.state = out.states[-1]
            self._update_ego(ep, out.planned_path)                    # 4. ego update
            ep._pending = (pred, out.planned_path)
        after = self._metrics(eps)                                    # 5. result metrics on the new ego state
        goal = self.engine.frenet_states([PlanRequest(ep.ego.x, ep.ego.y, ep.ego.yaw, ep.ego.v, ep.ego.a,
                                                      prev_s=ep.goal_prev_s) for ep in eps])[2]
        for ep, m, s_now in zip(eps, after, goal):
            pred, path = ep._pending
            stats = ep.pstate.last_check_stats
            if stats is not None:
                m["n_collision_rejected"] = stats.get("collision_error", 0)
            ep.history.append(StepRecord(ep.time, ep.ego, ep.peds.positions.copy(), ep.peds.current_velocities.copy(),
                                         ep.peds.goals.copy(), pred, path, m,
                                         {"prediction": t_pred, "planning": t_plan}))
            ep.time += self.dt
            ep.step_count += 1
            ep.goal_prev_s = float(s_now)
            if m["collision"]:                                        # run(): :864-883
                ep.termination_reason = "collision"
            elif self.s_end - float(s_now) < 2.0:
                ep.termination_reason = "goal"
        return len(eps)

    def run(self, n_steps: Optional[int] = None) -> List[List[StepRecord]]:
        if n_steps is None:
            n_steps = int(self.config.total_time / self.config.dt)
        for _ in range(n_steps):
            if self.step() == 0:
                break
        for ep in self.episodes:
            if ep.termination_reason is None:
                ep.termination_reason = "timeout"
        return [ep.history for ep in self.episodes]

    # ------------------------------------------------------------------------------------------------------
    @staticmethod
    def trajectory_arrays(history: List[StepRecord]) -> Dict[str, np.ndarray]:
        """The arrays of trajectory.npz (integrated_simulator.py:906-982): same keys, dtypes and shapes."""
        def planned(f):
            return np.array([np.array(getattr(r.planned_path, f)) if r.planned_path is not None else np.array([])
                             for r in history], dtype=object)
        return dict(
            times=np.array([r.time for r in history]),
            ego_x=np.array([r.ego.x for r in history]), ego_y=np.array([r.ego.y for r in history]),
            ego_v=np.array([r.ego.v for r in history]), ego_yaw=np.array([r.ego.yaw for r in history]),
            ego_jerk=np.array([r.ego.jerk for r in history]),
            ego_state=np.array([r.ego.state.name for r in history]),
            min_distances=np.array([r.metrics.get("min_distance", float("inf")) for r in history]),
            ttc=np.array([r.metrics.get("ttc", float("inf")) for r in history]),
            proc_prediction=np.array([r.processing_times.get("prediction", 0.0) for r in history]),
            proc_planning=np.array([r.processing_times.get("planning", 0.0) for r in history]),
            ped_positions=np.array([r.ped_positions for r in history], dtype=object),
            ped_velocities=np.array([r.ped_velocities for r in history], dtype=object),
            ped_goals=np.array([r.ped_goals for r in history], dtype=object),
            predicted_trajectories=np.array([r.predicted_trajectories if r.predicted_trajectories is not None
                                             else np.empty((0,)) for r in history], dtype=object),
            planned_x=planned("x"), planned_y=planned("y"), planned_v=planned("v"), planned_a=planned("a"),
            planned_yaw=planned("yaw"),
            planned_cost=np.array([r.planned_path.cost if r.planned_path is not None else float("inf")
                                   for r in history]))

    def save_results(self, output_path: str) -> List[str]:
        """One directory per episode (episode_000, ...), each with the reference's trajectory.npz."""
        files = []
        for i, ep in enumerate(self.episodes):
            d = os.path.join(output_path, f"episode_{i:03d}") if len(self.episodes) > 1 else output_path
            os.makedirs(d, exist_ok=True)
            f = os.path.join(d, "trajectory.npz")
            np.savez(f, **self.trajectory_arrays(ep.history))
            files.append(f)
        return files

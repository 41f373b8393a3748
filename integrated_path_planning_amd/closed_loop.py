"""Batched closed-loop driver: many simulator episodes in lock-step on one GPU.

One episode is what ``IntegratedSimulator.step()/run()`` (src/simulation/integrated_simulator.py:678-892) does for one
ego vehicle: advance the pedestrians, sample the observer, predict, prepend the current positions, safety metrics,
the escalate-and-retry planning cycle, ego update or emergency stop, termination on collision / goal / timeout.  Here
N episodes that share planner parameters and reference path advance together, and every step issues

* ONE constant-velocity prediction launch over the pedestrians of all running episodes (row f1),
* ONE safety-metrics launch before planning and one after the ego update (row f3),
* ONE ``fot_plan_batch`` with the current configuration of every episode and, only in steps where some first attempt
  fails, ONE more with every further escalation level of the failed episodes (row f2),
* ONE nearest-point launch for the goal test,

instead of N x (1 + up to 3 retries) sequential ``plan()`` calls.  All per-episode state (ego, state machine, planner
caches, pedestrian frame) lives in arrays; the reference's scalar control flow is restated as masked array updates, and
the histories are recorded as per-step arrays that turn into ``StepRecord`` objects only when read.  Pedestrians are
replayed tracks -- the contract of the reference's ``ReplayPedestrianSource`` (src/simulation/replay_source.py:31-118);
the Social-Force simulator and the Social-GAN network are outside SURVEY 8.  ``save_results`` writes
``trajectory.npz`` with the reference's keys, dtypes and array shapes (integrated_simulator.py:906-982), so the
existing analysis scripts read it unchanged.

All arithmetic on candidate paths, predictions and metrics runs in libfot.
"""
from __future__ import annotations

import math
import os
import time
from collections import deque
from dataclasses import dataclass, field
from typing import Any, Dict, List, Optional, Sequence

import numpy as np

from .batch import PlanRequest
from .data_structures import EgoVehicleState, FrenetPath, PedestrianState
from .footprint import EgoFootprint
from .planner import BatchPlanner
from .prediction import PredictionResampler
from . import _abi
from .state_machine import FailSafeStateMachine, VehicleState


class ReplayPedestrians:
    """Frame-by-frame replay of [T, N, 2] tracks: step()/get_state() of replay_source.py:31-118."""

    def __init__(self, trajectories, dt: float, velocities=None, goals=None, ids=None):
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
        self.ids = np.asarray(ids) if ids is not None else np.arange(self.n_peds)      # (replay_source.py:71-73)

    def step(self, ego_state=None, n: int = 1) -> None:
        """Advance n frames (the position holds at the last frame, the clock keeps running; the ego is ignored:
        replayed pedestrians do not react, replay_source.py:86-99)."""
        for _ in range(n):
            if self._idx < self.n_frames - 1:
                self._idx += 1
            self.time += self.dt

    def reset(self) -> None:
        self._idx, self.time = 0, 0.0

    @property
    def positions(self) -> np.ndarray:
        return self.trajectories[self._idx]

    @property
    def current_velocities(self) -> np.ndarray:
        return self.velocities[self._idx]

    def get_state(self) -> PedestrianState:
        """The current frame as the reference's carrier (replay_source.py:98-106): attribute access, ids included."""
        return PedestrianState(positions=self.positions.copy(), velocities=self.current_velocities.copy(),
                               goals=self.goals.copy(), ids=self.ids.copy(), timestamp=self.time)


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

    def reset(self) -> None:
        self.history.clear(); self.timestamps.clear()
        self.accumulated_time = 0.0
        self._last_update_timestamp = None

    @property
    def is_ready(self) -> bool:
        return len(self.history) >= self.obs_len

    @property
    def last_sample_time(self) -> Optional[float]:
        return self.timestamps[-1] if self.timestamps else None


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


_STATES = (VehicleState.NORMAL, VehicleState.CAUTION, VehicleState.EMERGENCY)      # array code 0, 1, 2


class _VectorStateMachine:
    """``FailSafeStateMachine`` (state_machine.py here, src/core/state_machine.py:29-278 in the reference) for all
    episodes at once: the same transitions and planner configurations, as masked array updates.  Codes 0 / 1 / 2 =
    NORMAL / CAUTION / EMERGENCY."""

    def __init__(self, config, n: int):
        one = FailSafeStateMachine(config)                       # the scalar class resolves the configuration keys
        c = config
        self.clr_caution, self.clr_emergency = one.clearance_caution, one.clearance_emergency
        self.trig_c, self.trig_h = one.trigger_clearance_caution, one.trigger_time_headway
        self.env_decel, self.env_standoff = one.envelope_decel, one.envelope_standoff
        self.target = float(c.ego_target_speed)
        self.c_accel = c.ego_max_accel * getattr(c, "state_machine_caution_accel_multiplier", 1.5)
        self.c_speed_mult = getattr(c, "state_machine_caution_speed_multiplier", 0.8)
        self.c_speed = c.ego_max_speed * self.c_speed_mult
        self.e_accel = c.ego_max_accel * getattr(c, "state_machine_emergency_accel_multiplier", 3.0)
        self.e_lat = getattr(c, "ego_max_lat_accel", 3.0) * getattr(c, "state_machine_emergency_lat_accel_multiplier", 2.0)
        self.state = np.zeros(n, np.int64)
        self.fails = np.zeros(n, np.int64)
        self.clear = np.full(n, np.inf)                          # _last_clearance
        self.clear_ahead = np.full(n, np.inf)                    # _last_clearance_ahead

    def config(self, state: np.ndarray, clear_ahead: np.ndarray):
        """_get_planner_config (:181-247) -> target speed, overrides [n, 4] (NaN = absent), max_stop (NaN = None)."""
        n = len(state)
        fin = np.isfinite(clear_ahead)
        has_env = fin & (self.env_decel > 0.0)
        v_env = np.sqrt(2.0 * self.env_decel * np.maximum(np.where(fin, clear_ahead, 0.0) - self.env_standoff, 0.0))
        stop_room = np.where(fin, np.maximum(np.where(fin, clear_ahead, 0.0) - 0.2, 0.05), np.nan)
        target = np.full(n, self.target)
        ov = np.full((n, 4), np.nan)
        stop = np.full(n, np.nan)
        nm, ca, em = state == 0, state == 1, state == 2
        target = np.where(nm & has_env & (v_env < self.target), v_env, target)
        t_ca = np.where(has_env, np.minimum(self.target * self.c_speed_mult, v_env), self.target * self.c_speed_mult)
        target = np.where(ca, t_ca, target)
        stop = np.where(ca & has_env & (v_env <= 0.0), stop_room, stop)
        ov[ca, 1], ov[ca, 0] = self.c_accel, self.c_speed
        target = np.where(em, 0.0, target)
        ov[em, 1], ov[em, 3] = self.e_accel, self.e_lat
        if self.env_decel > 0.0:
            stop = np.where(em, stop_room, stop)
        return target, ov, stop

    def update(self, sel: np.ndarray, found: np.ndarray, clearance: np.ndarray, clearance_ahead: np.ndarray,
               speed: np.ndarray) -> None:
        """update() (:116-179) for the episodes ``sel`` (index array): observe the metrics, then the transitions."""
        self.clear[sel], self.clear_ahead[sel] = clearance, clearance_ahead
        st, fl = self.state[sel], self.fails[sel]
        trigger = self.trig_c + self.trig_h * np.maximum(speed, 0.0)
        nm, ca, em = st == 0, st == 1, st == 2
        new_st, new_fl = st.copy(), fl.copy()
        a = nm & ~found
        new_st[a] = 1; new_fl[a] = fl[a] + 1
        b = nm & found & (trigger > 0.0) & (clearance < trigger)
        new_st[b] = 1; new_fl[b] = 0
        new_fl[nm & found & ~b] = 0
        c1 = ca & found & (fl == 0)
        new_st[c1 & (clearance > np.maximum(self.clr_caution, trigger))] = 0
        c2 = ca & ~c1 & ~found
        new_st[c2] = 2; new_fl[c2] = fl[c2] + 1
        new_fl[ca & ~c1 & found] = 0
        new_st[em & found & (clearance > self.clr_emergency)] = 1
        self.state[sel], self.fails[sel] = new_st, new_fl


class EpisodeHistory:
    """One episode's steps as a read-only sequence of ``StepRecord`` built on demand from the loop's per-step arrays
    (every episode takes part in every lock step from the first one until it ends: its step i is lock step i)."""

    def __init__(self, loop: "BatchedClosedLoop", e: int):
        self._loop, self._e = loop, e

    def __len__(self):
        return int(self._loop.step_counts[self._e])

    def __iter__(self):
        return (self[i] for i in range(len(self)))

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self[k] for k in range(*i.indices(len(self)))]
        if i < 0:
            i += len(self)
        if not 0 <= i < len(self):
            raise IndexError(i)
        return self._loop._record(i, self._e)


_TERMINATION = (None, "collision", "goal", "timeout")


class Episode:
    """View of one episode of the loop: its history, how many steps it ran and why it ended (None while it runs)."""

    def __init__(self, loop: "BatchedClosedLoop", e: int):
        self._loop, self._e = loop, e
        self.history = EpisodeHistory(loop, e)

    @property
    def step_count(self) -> int:
        return int(self._loop.step_counts[self._e])

    @property
    def termination_reason(self) -> Optional[str]:
        return _TERMINATION[int(self._loop.termination[self._e])]

    @termination_reason.setter
    def termination_reason(self, reason: Optional[str]) -> None:
        self._loop.termination[self._e] = _TERMINATION.index(reason)


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


def emergency_stop(x, y, yaw, v, clearance_ahead, dt, max_accel, emergency_decel=None):
    """``IntegratedSimulator._apply_emergency_stop`` (integrated_simulator.py:749-802) for arrays of egos: the position
    integrates along the heading at the OLD speed, the deceleration is what stopping 0.2 m short of the nearest
    pedestrian ahead needs (v^2 / (2 max(clearance - 0.2, 0.05))), bounded to [max_accel, emergency_decel]
    (``None`` = 2 x max_accel); with nothing ahead (non-finite clearance) the cap itself.  Returns the new x, y, v, a
    (a = 0 once the vehicle stands)."""
    x, y, yaw, v = (np.asarray(q, dtype=float) for q in (x, y, yaw, v))
    clr = np.asarray(clearance_ahead, dtype=float)
    cap = max_accel * 2.0 if emergency_decel is None else emergency_decel
    fin = np.isfinite(clr)
    required = np.where(fin, v ** 2 / (2.0 * np.maximum(np.where(fin, clr, 1.0) - 0.2, 0.05)), cap)
    max_dec = np.clip(required, max_accel, cap)
    nv = np.maximum(0.0, v - max_dec * dt)
    na = np.where(nv > 0, -max_dec, 0.0)
    # (the C library's cos / sin element by element -- what math.cos calls and what fot_loop_step's C++ calls: NumPy's
    #  array loops are a SIMD routine of their own that may differ from it in the last place)
    cy = np.array([math.cos(float(t)) for t in np.atleast_1d(yaw)]).reshape(np.shape(yaw))
    sy = np.array([math.sin(float(t)) for t in np.atleast_1d(yaw)]).reshape(np.shape(yaw))
    return x + v * cy * dt, y + v * sy * dt, nv, na


class BatchedClosedLoop:
    """N episodes of the reference's closed loop in lock-step.

    config: the scenario dictionary (or an object with the same attributes) the reference's SimulationConfig is
    built from; ped_tracks: one [T, N_i, 2] array of replayed pedestrian positions per episode (frame spacing
    config.dt, frame 0 = time 0 before warm-up); ego_initial_states: optional per-episode [x, y, yaw, v, a].

    The state of all episodes lives in arrays (ego, state machine, planner caches, pedestrian frames); a lock step is a
    fixed sequence of array operations and five libfot calls, whatever the number of episodes.  Histories are recorded
    as per-step arrays and turned into ``StepRecord`` objects only when somebody reads them.
    """

    MAX_REPLAN = 3                                               # integrated_simulator.py:383

    def __init__(self, config, ped_tracks: Sequence[np.ndarray], ego_initial_states: Optional[Sequence] = None,
                 device: int = -1, engine=None, resampler=None, sample_source=None, fused: Optional[bool] = None,
                 device_samples: bool = False):
        """sample_source: the multi-sample predictor in front of the planner -- a callable
        ``(obs_last [P, 2], obs_prev [P, 2]) -> raw samples [S, pred_len, P, 2]`` at the predictor's own time step
        (what S forward passes of Social-GAN on PyTorch-ROCm return for the pedestrians of all running episodes; the
        tests script one).  With it the episodes plan against the whole distribution when the configuration says
        ``distribution_aware_planning`` (integrated_simulator.py:459-460, 514-525), otherwise against the sample closest
        to the mean (``predict_single_best``, trajectory_predictor.py:340-352).  None: the constant-velocity predictor.
        device_samples: the sample source returns a ``torch`` tensor in DEVICE memory ([S, pred_len, sum P, 2], float32 or
        float64) -- Social-GAN's own output on PyTorch-ROCm.  With ``distribution_aware_planning`` the samples then never
        leave the GPU: they are resampled into the planner's tensor inside the lock step's one call (fot_loop_step)."""
        self.config = config if not isinstance(config, dict) else _Cfg(config)
        c = self.config
        self.dt = float(c.dt)
        self.ego_radius = getattr(c, "ego_radius", 1.0)
        self.ped_radius = getattr(c, "ped_radius", 0.3)
        self.footprint = footprint_from_config(c)
        self.sample_source = sample_source
        if sample_source is None and getattr(c, "prediction_method", "sgan") != "cv":
            raise NotImplementedError("the Social-GAN / LSTM networks are not part of this build (SURVEY 8 f1): hand "
                                      "their samples in through sample_source, or use prediction_method='cv'")
        self.distribution_aware = bool(getattr(c, "distribution_aware_planning", False))
        if self.distribution_aware and sample_source is None:
            raise ValueError("distribution_aware_planning needs a multi-sample predictor (sample_source): the "
                             "constant-velocity predictor yields one sample")
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
        # the step's device work in two calls, prediction resident in HBM (fot_loop_*): the constant-velocity predictor
        # on the library's own engine; a sample source hands its samples over on the host, stand-in engines have no device
        self._device_samples = bool(device_samples)
        if self._device_samples and not (sample_source is not None and self.distribution_aware and engine is None and resampler is None):
            raise ValueError("device_samples needs a sample_source, distribution_aware_planning and the library's own engine")
        can_fuse = (sample_source is None or self._device_samples) and resampler is None and hasattr(self.engine, "loop_plan")
        if fused not in (None, False, True, "two-call"):
            raise ValueError("fused: None (automatic), False, True or 'two-call'")
        if fused and not can_fuse:
            raise ValueError("fused=True needs the constant-velocity predictor on the library's own engine")
        self._fused = can_fuse if fused is None else bool(fused)
        # ... and, on the library's own engine, the whole step behind ONE call (fot_loop_step: the episodes' state, the
        # fail-safe machine and the retry loop live in the handle); fused="two-call" keeps the round-3 form (two calls,
        # the retry loop replayed here on arrays) -- the tests run both against each other and against the five-call step
        self._native = self._fused and fused != "two-call" and hasattr(self.engine, "loop_step")
        if self._device_samples and not self._native:
            raise ValueError("device_samples runs through the one-call step only")
        if self._fused:
            self.engine.loop_set_static(self.static_obstacle_points)
        self.sgan_dt = 0.4                                            # integrated_simulator.py:323-327
        self.resampler = resampler if resampler is not None else PredictionResampler(
            self.engine, pred_len=c.pred_len, sgan_dt=self.sgan_dt, sim_dt=c.dt, plan_horizon=getattr(c, "max_t", 5.0))
        n = len(ped_tracks)
        if ego_initial_states is None:
            ego_initial_states = [c.ego_initial_state] * n
        # ---- pedestrians: replayed tracks, every episode on the same clock (replay_source.py:31-118)
        self.peds = [ReplayPedestrians(tr, c.dt) for tr in ped_tracks]
        self.n_frames = np.array([p.n_frames for p in self.peds])
        self.ped_off = np.concatenate([[0], np.cumsum([p.n_peds for p in self.peds])]).astype(np.int64)
        # all episodes' tracks side by side, [T_max, sum P, 2] (a shorter replay holds its last frame, as step() does):
        # a frame of every running episode is then one row selection instead of a Python loop over the episodes
        t_max = int(self.n_frames.max()) if len(self.peds) else 0

        def side_by_side(which):
            out = np.zeros((t_max, int(self.ped_off[-1]), 2))
            for e, pd_ in enumerate(self.peds):
                a = getattr(pd_, which)
                out[: pd_.n_frames, self.ped_off[e]:self.ped_off[e + 1]] = a
                out[pd_.n_frames:, self.ped_off[e]:self.ped_off[e + 1]] = a[-1]
            return out
        self._ped_all = {"trajectories": side_by_side("trajectories"), "velocities": side_by_side("velocities")}
        self._rows_key, self._rows = None, None
        self.frame, self.ped_time = 0, 0.0
        self.observer = Observer(c.obs_len, c.dt, self.sgan_dt)      # one sampling clock; samples = all episodes' peds
        # ---- ego, state machine and planner caches as arrays
        e0 = np.array([np.asarray(v, float)[:5] for v in ego_initial_states], dtype=float).reshape(n, 5)
        self.ego = e0.copy()                                         # x, y, yaw, v, a
        self.jerk = np.array([float(np.asarray(v, float)[5]) if len(v) > 5 else 0.0 for v in ego_initial_states])
        self.sm = _VectorStateMachine(c, n)
        self.prev_s = np.full(n, np.nan)                             # planner.converter._prev_s (NaN: not set yet)
        self.last_kappa = np.zeros(n)                                # planner._last_kappa
        self.goal_prev_s = np.full(n, np.nan)                        # the simulator's own converter (:873)
        self.last_clearance = np.full(n, np.inf)
        self.last_stats = np.full((n, 8), -1, np.int64)              # last_check_stats (-1 row: None)
        self.time = 0.0
        self.alive = np.ones(n, bool)
        self._steps: List[dict] = []
        # Where the selected paths of every step are kept (15 arrays x episodes x samples per step): chunks of whole steps,
        # touched when they are allocated -- fresh pages cost the step that first writes them about as much as the device
        # work of the whole step.  The first chunk covers the configured duration.
        self._arena: List[np.ndarray] = []
        self._arena_steps = 0
        self._arena_slot = len(_abi.PATH_FIELDS) * max(n, 1) * int(getattr(self.engine, "n_total_samples", _abi.MAX_NT))
        if hasattr(self.engine, "gather_paths"):
            self._grow_arena(int(getattr(c, "total_time", 0.0) / self.dt) + 1)
        self.step_counts = np.zeros(n, np.int64)                     # lock steps each episode took part in
        self.termination = np.zeros(n, np.int8)                      # index into _TERMINATION
        self.episodes: List[Episode] = [Episode(self, e) for e in range(n)]
        self._warmup()
        if self._native:
            sm_, lc = self.sm, _abi.LoopConfig()
            lc.dt, lc.target_speed, lc.max_accel = float(c.dt), float(sm_.target), float(c.ego_max_accel)
            dec = getattr(c, "ego_emergency_decel", None)
            lc.emergency_decel = float("nan") if dec is None else float(dec)
            lc.clearance_caution, lc.clearance_emergency = float(sm_.clr_caution), float(sm_.clr_emergency)
            lc.trigger_clearance_caution, lc.trigger_time_headway = float(sm_.trig_c), float(sm_.trig_h)
            lc.envelope_decel, lc.envelope_standoff = float(sm_.env_decel), float(sm_.env_standoff)
            lc.caution_accel, lc.caution_speed, lc.caution_speed_mult = float(sm_.c_accel), float(sm_.c_speed), float(sm_.c_speed_mult)
            lc.emergency_accel, lc.emergency_lat_accel = float(sm_.e_accel), float(sm_.e_lat)
            lc.max_replan = self.MAX_REPLAN
            self.engine.loop_begin(lc, self.ego)

    def close(self) -> None:
        """Release the libfot handle (streams, workspace) now rather than at garbage collection."""
        if self.engine is not None:
            for s in self._steps:                                     # predictions of fused steps nobody has read yet
                if s["pred"] is None and s.get("pred_src") is not None and not isinstance(s["pred_src"][0], str):
                    s["pred"] = self._materialise_prediction(s["pred_src"], s["off"])
                    s["pred_src"] = None                              # (a distribution's samples: only while somebody asks)
        if self._owns_engine and self.engine is not None:
            self.engine.close()
        self.engine = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # ------------------------------------------------------------------------------------------------------
    ARENA_CHUNK_BYTES = 1 << 28

    def _grow_arena(self, steps: int) -> None:
        steps = max(1, min(int(steps), self.ARENA_CHUNK_BYTES // (8 * self._arena_slot) or 1))
        chunk = np.empty((steps, self._arena_slot))
        chunk.fill(0.0)                                               # (touch the pages now, not inside a step)
        self._arena.append(chunk)
        self._arena_steps += steps

    def _history_block(self, k: int, n: int, kmax: int) -> np.ndarray:
        """[15, n, kmax] block of lock step k in the arena."""
        while k >= self._arena_steps:
            self._grow_arena(64)
        for chunk in self._arena:
            if k < len(chunk):
                return chunk[k, : len(_abi.PATH_FIELDS) * n * kmax].reshape(len(_abi.PATH_FIELDS), n, kmax)
            k -= len(chunk)
        raise IndexError(k)

    def _ped_frame(self, which: str, sel: np.ndarray) -> np.ndarray:
        """positions / velocities of the episodes ``sel`` at the current frame, concatenated [sum P, 2]."""
        frame = self._ped_all[which][min(self.frame, len(self._ped_all[which]) - 1)]
        return frame if len(sel) == len(self.peds) else frame[self._rows_of(sel)]

    def _rows_of(self, sel: np.ndarray) -> np.ndarray:
        """Pedestrian rows of the episodes ``sel`` (cached: the set of running episodes changes rarely)."""
        key = sel.tobytes()
        if key != self._rows_key:
            self._rows_key = key
            self._rows = (np.concatenate([np.arange(self.ped_off[e], self.ped_off[e + 1]) for e in sel])
                          if len(sel) else np.zeros(0, np.int64))
        return self._rows

    def _advance_pedestrians(self) -> None:
        self.frame += 1
        self.ped_time += self.dt
        every = np.arange(len(self.peds))
        self.observer.update(self._ped_frame("trajectories", every), self.ped_time)

    def _warmup(self) -> None:
        """integrated_simulator.py:406-422: fill the observers before time 0."""
        c = self.config
        for _ in range(int(c.obs_len * self.sgan_dt / c.dt)):
            self._advance_pedestrians()

    @property
    def running(self) -> List[Episode]:
        return [ep for ep, a in zip(self.episodes, self.alive) if a]

    def _metrics(self, sel, off, pos, vel):
        egos = np.stack([self.ego[sel, 0], self.ego[sel, 1], self.ego[sel, 2], self.ego[sel, 3]], axis=1)
        return self.engine.safety_metrics_cat(egos, off, pos, vel, self.ego_radius, self.ped_radius,
                                              use_footprint=self.footprint is not None)

    def _predict(self, sel, off, pos):
        """_update_prediction (:424-527): one launch over the pedestrians of all running episodes.  Returns the
        prediction [sum P, T, 2] (None while the observer fills), per episode whether the current positions are
        prepended (:503-511), and the whole distribution [S, sum P, T, 2] (None unless a sample source predicts)."""
        t0 = time.perf_counter()
        pred, dist = None, None
        if self.observer.is_ready:
            rows = self._rows_of(sel)
            hist = self.observer.history
            obs = np.stack([hist[-2][rows], hist[-1][rows]], axis=0)          # the last two samples
            last = self.observer.last_sample_time
            stale = max(self.ped_time - last, 0.0) if last is not None else 0.0
            if self.sample_source is None:
                pred = self.resampler.predict_cv(obs, staleness=stale, float32_observations=True)
            else:
                # the observer hands over float32 tensors (observer.py:134); the samples are resampled to the
                # simulation step on the device (process_prediction, :233-313), all pedestrians in one launch
                o32 = obs.astype(np.float32).astype(np.float64)
                raw = np.asarray(self.sample_source(o32[1], o32[0]), dtype=np.float64)       # [S, pred_len, sum P, 2]
                dist = self.resampler.process_prediction(raw, anchor_pos=o32[1], staleness=stale)
                if raw.shape[0] == 1:
                    pred, dist = dist[0], None
                else:
                    # predict_single_best (:340-352), per episode: the sample closest to the sample mean over the
                    # episode's own pedestrians
                    pred = self._best_sample(dist, off)
        t_pred = (time.perf_counter() - t0) / len(sel)
        if pred is None:
            return None, np.zeros(len(sel), bool), t_pred, None
        # np.allclose(pred[:, 0], current) of the reference (rtol 1e-5, atol 1e-8; finite inputs), per episode
        close = np.all(np.abs(pred[:, 0, :] - pos) <= 1e-8 + 1e-5 * np.abs(pos), axis=1)
        same = np.logical_and.reduceat(close, off[:-1]) if len(close) else np.zeros(len(sel), bool)
        return pred, ~same, t_pred, dist if self.distribution_aware else None

    # ------------------------------------------------------------------------------------------------------
    def step(self) -> int:
        """One lock step of every running episode (integrated_simulator.py:678-747); returns how many ran."""
        sel = np.flatnonzero(self.alive)
        n = len(sel)
        if n == 0:
            return 0
        c, sm = self.config, self.sm
        self._advance_pedestrians()                                   # 1. pedestrians + observer
        counts = (self.ped_off[sel + 1] - self.ped_off[sel]).astype(np.int64)
        off = np.concatenate([[0], np.cumsum(counts)])
        pos = self._ped_frame("trajectories", sel)
        vel = self._ped_frame("velocities", sel)
        st0 = sm.state[sel]
        n_lvl = np.minimum(3 - st0, 1 + self.MAX_REPLAN)             # NORMAL -> CAUTION -> EMERGENCY, then no change
        everyone = np.arange(n)
        speed = self.ego[sel, 3].copy()
        if self._native:
            return self._step_native(sel, off, pos, vel)
        if self._fused:
            return self._step_fused(sel, off, counts, pos, vel, st0, n_lvl, everyone, speed)
        pred, prepend, t_pred, dist = self._predict(sel, off, pos)    # 2. prediction
        m = self._metrics(sel, off, pos, vel)                         # 3. planning cycle (:529-653)
        t0 = time.perf_counter()
        clearance, clearance_ahead = m["clearance"].copy(), m["clearance_ahead"].copy()
        self.last_clearance[sel] = clearance_ahead
        # --- level 0 of every episode = the current state's configuration (issued from LAST step's clearance)
        # --- obstacles: the same static points for every request; one dynamic tensor per episode, shared by its levels
        pts = self.static_obstacle_points
        if pred is None:                                              # not ready: current positions only (:495-498)
            dyn, t_len = pos[:, None, :], np.ones(n, np.int64)
        elif prepend.all():
            dyn, t_len = np.concatenate([pos[:, None, :], pred], axis=1), np.full(n, pred.shape[1] + 1, np.int64)
        elif not prepend.any():
            dyn, t_len = pred, np.full(n, pred.shape[1], np.int64)
        else:                                                         # mixed: the shorter tensors end one sample early
            T1 = pred.shape[1] + 1
            dyn = np.concatenate([pos[:, None, :], pred], axis=1)
            ped_pre = np.repeat(prepend, counts)
            dyn[~ped_pre, :-1] = pred[~ped_pre]
            t_len = np.where(prepend, T1, T1 - 1)
        T_alloc = dyn.shape[1]
        n_smp = np.ones(n, np.int64)
        mode = np.where(counts > 0, 1, 0)
        if dist is not None:
            # the planner consumes the whole distribution (:622-630); the current positions lead EVERY sample,
            # whatever the single sample's prepend decided (:514-525): per episode a [S, P, T + 1, 2] block
            S_ = dist.shape[0]
            full = np.concatenate([np.broadcast_to(pos[None, :, None, :], (S_, len(pos), 1, 2)), dist], axis=2)
            blocks = [np.ascontiguousarray(full[:, off[i]:off[i + 1]]).reshape(-1, 2) for i in range(n)]
            d_xy = np.concatenate(blocks, axis=0) if blocks else np.empty((0, 2))
            t_len = np.full(n, full.shape[2], np.int64)
            d_off_ep = np.concatenate([[0], np.cumsum(S_ * counts * t_len)])[:-1]
            n_smp = np.full(n, S_, np.int64)
            mode = np.where(counts > 0, 2, 0)
        elif t_len.min() != T_alloc:                                  # mixed case: per-episode [P, t_len, 2] blocks
            blocks = [np.ascontiguousarray(dyn[off[i]:off[i + 1], :t_len[i]]).reshape(-1, 2) for i in range(n)]
            d_xy = np.concatenate(blocks, axis=0)
            d_off_ep = np.concatenate([[0], np.cumsum(counts * t_len)])[:-1]
        else:
            d_xy = np.ascontiguousarray(dyn).reshape(-1, 2)
            d_off_ep = off[:-1] * T_alloc

        def plan(who, state, clear_ahead, prev_s, chain):
            """one plan() per entry: episode who[i] under the configuration of `state[i]`; chain[i]: nearest-point
            cache handed over from the entry before (the next escalation level of the same episode)"""
            tgt, ov, stop = sm.config(state, clear_ahead)
            r = len(who)
            ego = np.zeros(r, dtype=self.engine.EGO_DT)
            for col, f in enumerate(("x", "y", "yaw", "v", "a")):
                ego[f] = self.ego[sel, col][who]
            ego["last_kappa"] = self.last_kappa[sel][who]
            ego["has_prev_s"] = np.where(chain, 2, ~np.isnan(prev_s))
            ego["prev_s"] = np.where(chain | np.isnan(prev_s), 0.0, prev_s)
            s_xy = np.tile(pts, (r, 1)) if len(pts) else None
            s_off = np.arange(r + 1, dtype=np.int64) * len(pts) if len(pts) else None
            d_dims = np.stack([mode[who], n_smp[who], counts[who], t_len[who]], axis=1)
            return self.engine.plan_arrays(ego, tgt, ov, stop, s_xy, s_off, d_xy, d_off_ep[who], d_dims)

        return self._finish_step(sel, off, pos, vel, pred, None, t_pred, t0, plan, st0, n_lvl, everyone, speed, clearance,
                                 clearance_ahead, lambda new_ego: (lambda r=(self._metrics(sel, off, pos, vel), self.engine.nearest_s_arrays(
                                     new_ego[:, 0], new_ego[:, 1], new_ego[:, 2], new_ego[:, 3], new_ego[:, 4],
                                     self.goal_prev_s[sel])): r))

    def _step_fused(self, sel, off, counts, pos, vel, st0, n_lvl, everyone, speed):
        """Steps 2-5 with the device work in two libfot calls (fot_loop_plan / fot_loop_observe): prediction, current
        metrics and the level-0 plans in one enqueue -- the prediction tensor is written and read in HBM --, the new
        state's metrics and the goal test's nearest point in another.  The step record keeps the observer's two samples
        instead of the prediction, which is computed again (same kernel, same numbers) if somebody reads it."""
        sm = self.sm
        n = len(sel)
        t0 = time.perf_counter()
        frame, pred_src = self._loop_frame(sel, off, pos, vel)

        def requests(who, state, clear_ahead, prev_s, chain):
            # fot_loop_request is 15 eight-byte slots: x y yaw v a last_kappa prev_s | has_prev_s, pad | 4 overrides |
            # target_speed max_stop_distance | episode, pad -- filled column-wise through a float64 / int32 view
            tgt, ov, stop = sm.config(state, clear_ahead)
            req = np.zeros(len(who), dtype=self.engine.LOOP_REQUEST_DT)
            f64 = req.view(np.float64).reshape(len(who), 15)
            i32 = req.view(np.int32).reshape(len(who), 30)
            f64[:, 0:5] = self.ego[sel[who]]
            f64[:, 5] = self.last_kappa[sel[who]]
            f64[:, 6] = np.where(chain | np.isnan(prev_s), 0.0, prev_s)
            i32[:, 14] = np.where(chain, 2, ~np.isnan(prev_s))
            f64[:, 8:12] = ov
            f64[:, 12], f64[:, 13] = tgt, stop
            i32[:, 28] = who
            return req

        # (view=True: the records are read -- the selected paths copied into the history arena -- before the next call)
        rec0, m = self.engine.loop_plan(requests(everyone, st0, sm.clear_ahead[sel], self.prev_s[sel], np.zeros(n, bool)),
                                        frame, view=True)
        t_pred = 0.0                                                  # (inside the one call: not separable)
        clearance, clearance_ahead = m["clearance"].copy(), m["clearance_ahead"].copy()
        self.last_clearance[sel] = clearance_ahead
        plan = lambda *a: self.engine.loop_plan(requests(*a), view=True)[0]
        return self._finish_step(sel, off, pos, vel, None, pred_src, t_pred, t0, plan, st0, n_lvl, everyone, speed,
                                 clearance, clearance_ahead, lambda new_ego: self.engine.loop_observe_begin(new_ego, self.goal_prev_s[sel]),
                                 first=rec0)

    def _best_sample(self, dist: np.ndarray, off: np.ndarray) -> np.ndarray:
        """predict_single_best (trajectory_predictor.py:340-352) per episode: the sample closest to the sample mean over
        the episode's own pedestrians -> [sum P, T, 2]."""
        n = len(off) - 1
        dev = np.linalg.norm(dist - dist.mean(axis=0)[None], axis=-1).sum(axis=2)      # [S, sum P]
        per_ep = np.add.reduceat(dev, off[:-1], axis=1) if dev.shape[1] else np.zeros((len(dist), n))
        per_ep[:, off[:-1] == off[1:]] = 0.0                                      # (episodes without pedestrians)
        best = np.argmin(per_ep, axis=0)                                         # [episodes]
        return dist[np.repeat(best, off[1:] - off[:-1]), np.arange(dist.shape[1])]

    def _materialise_prediction(self, pred_src, off):
        """The prediction a fused step left in HBM, computed again for whoever reads the step's record (same kernels,
        same numbers): the constant-velocity tracks, or the best sample of the distribution's raw samples."""
        if isinstance(pred_src[0], str):                              # ("dist", raw samples in HBM, observations, staleness)
            _, raw, o32, stale = pred_src
            raw_h = raw.detach().cpu().numpy().astype(np.float64)
            dist = self.resampler.process_prediction(raw_h, anchor_pos=o32[1].astype(np.float64), staleness=stale)
            return dist[0] if raw_h.shape[0] == 1 else self._best_sample(dist, np.asarray(off))
        o32, stale = pred_src
        return self.resampler.predict_cv(o32, staleness=stale, float32_observations=True)

    def _loop_frame(self, sel, off, pos, vel):
        """The frame of fot_loop_plan / fot_loop_step for the running episodes: pedestrians, the observer's last two
        samples, per episode whether the current positions lead the prediction (:503-511), staleness."""
        frame = dict(ped_off=off, ped_pos=pos, ped_vel=vel, ego=self.ego[sel, :4], ego_radius=self.ego_radius,
                     ped_radius=self.ped_radius, use_footprint=self.footprint is not None)
        pred_src = None
        if self.observer.is_ready:
            hist = self.observer.history
            if len(sel) == len(self.peds):                            # every episode still runs: the samples as they are
                o32 = np.empty((2,) + hist[-1].shape, np.float32)
                o32[0], o32[1] = hist[-2], hist[-1]
            else:
                rows = self._rows_of(sel)
                o32 = np.stack([hist[-2][rows], hist[-1][rows]], axis=0).astype(np.float32)
            last = self.observer.last_sample_time
            stale = max(self.ped_time - last, 0.0) if last is not None else 0.0
            if self._device_samples:
                prepend = np.ones(len(sel), bool)                     # (the current positions lead EVERY sample, :514-525)
            else:
                # np.allclose(pred[:, 0], current) (:503-511) needs the first predicted sample only: obs_last + v (dt + stale),
                # the velocity formed in float32 as the kernel (and the reference, trajectory_predictor.py:216) forms it
                vel32 = (o32[1] - o32[0]) / np.float32(self.sgan_dt)
                first = o32[1].astype(np.float64) + vel32.astype(np.float64) * ((self.dt + 0.0 * self.dt) + stale)
                far = np.any(np.abs(first - pos) > 1e-8 + 1e-5 * np.abs(pos), axis=1)
                n_far = np.concatenate([[0], np.cumsum(far)])
                prepend = n_far[off[1:]] != n_far[off[:-1]]           # per episode (False without pedestrians)
            frame.update(obs_last=o32[1], obs_prev=o32[0], prepend=prepend, staleness=stale,
                         pred_len=self.resampler.pred_len, rp=self.resampler.params)
            pred_src = (o32, stale)
            if self._device_samples:
                # the multi-sample predictor's raw output stays in HBM: handed to the step as a device pointer
                raw = self.sample_source(o32[1].astype(np.float64), o32[0].astype(np.float64))
                if not (hasattr(raw, "data_ptr") and raw.is_cuda and raw.is_contiguous() and raw.dim() == 4):
                    raise TypeError("device_samples: the sample source must return a contiguous CUDA tensor [S, pred_len, sum P, 2]")
                import torch
                if raw.dtype not in (torch.float32, torch.float64):
                    raise TypeError("device_samples: float32 or float64 samples")
                torch.cuda.current_stream(raw.device).synchronize()   # (the library reads it on its own stream)
                frame.update(dist_raw=raw.data_ptr(), dist_S=int(raw.shape[0]),
                             dist_dtype=_abi.F32 if raw.dtype == torch.float32 else _abi.F64, _raw=raw)
                pred_src = ("dist", raw, o32, stale)

        return frame, pred_src

    def _step_native(self, sel, off, pos, vel):
        """Steps 2-5 behind ONE libfot call (fot_loop_step): the episodes' state -- ego, planner caches, fail-safe machine
        -- lives in the handle, the retry loop is replayed there; what is left here is the pedestrian frame, the
        observer, the history and the termination test."""
        n = len(sel)
        t0 = time.perf_counter()
        frame, pred_src = self._loop_frame(sel, off, pos, vel)
        frame.pop("ego")
        o = self.engine.loop_step(frame, sel)
        t_plan = (time.perf_counter() - t0) / n
        rec, path_rec, keep = o["records"], o["record"].astype(np.int64), o["keep"].astype(np.int64)
        new_ego = o["ego"]
        self.ego[sel], self.jerk[sel] = new_ego, o["jerk"]
        self.sm.state[sel] = o["state"]
        self.last_stats[sel] = o["stats"]
        chosen = np.maximum(path_rec, 0)
        kmax = int(keep.max()) if n else 0
        slot = np.full(len(self.episodes), -1, np.int64)
        slot[sel] = np.arange(n)
        block = self.engine.gather_paths(rec, chosen, kmax, out=self._history_block(len(self._steps), n, kmax))
        paths = {f: block[j] for j, f in enumerate(_abi.PATH_FIELDS)}
        after, s_now = o["after"], o["s_now"]
        self._steps.append(dict(
            time=self.time, slot=slot, off=off, ego=new_ego, jerk=o["jerk"], state=self.sm.state[sel].copy(), pos=pos, vel=vel,
            pred=None, pred_src=pred_src, after=after, stats=self.last_stats[sel].copy(), has_path=path_rec >= 0, keep=keep,
            cost=o["cost"], paths=paths, t_pred=0.0, t_plan=t_plan, sel=sel))
        collided = after["collision"] != 0
        at_goal = self.s_end - s_now < 2.0
        self.step_counts[sel] += 1
        self.termination[sel[at_goal & ~collided]] = 2
        self.termination[sel[collided]] = 1
        self.alive[sel[collided | at_goal]] = False
        self.time += self.dt
        return n

    def _finish_step(self, sel, off, pos, vel, pred, pred_src, t_pred, t0, plan, st0, n_lvl, everyone, speed, clearance,
                     clearance_ahead, observe, first=None):
        """The rest of a lock step, whoever planned: replay of the retry loop, ego update, result metrics, history.
        ``plan(who, state, clear_ahead, prev_s, chain)`` -> records; ``observe(new_ego)`` -> a callable that returns
        (metrics, nearest s) -- the fused step enqueues the two launches and collects them behind the bookkeeping;
        ``first``: the records of level 0 when they were planned already (with the frame's prediction and metrics)."""
        c, sm = self.config, self.sm
        n = len(sel)
        rec = first if first is not None else plan(everyone, st0, sm.clear_ahead[sel], self.prev_s[sel], np.zeros(n, bool))
        # --- replay of the retry loop (:576-653).  Episodes whose first attempt failed get every further escalation
        #     level they can reach planned in ONE more launch (the configurations update(False, ...) would issue on THIS
        #     step's metrics, nearest-point cache chained from attempt to attempt); the control flow is then replayed.
        found_all = rec["status"] == 0
        cur = everyone.copy()                                         # record of each episode's current attempt
        path_rec = np.full(n, -1, np.int64)
        failed = np.flatnonzero(~found_all[:n] & (n_lvl > 1))
        if len(failed):
            extra = n_lvl[failed] - 1
            who = np.repeat(failed, extra)
            base1 = np.concatenate([[0], np.cumsum(extra)])[:-1]
            lvl = 1 + np.arange(len(who)) - np.repeat(base1, extra)
            nps0 = rec["new_prev_s"][who]
            head = rec.copy()                                         # (a view of the handle's block when fused)
            rec = np.concatenate([head, plan(who, st0[who] + lvl, clearance_ahead[who],
                                             np.where(np.isnan(nps0), self.prev_s[sel][who], nps0), lvl > 1)])
            found_all = rec["status"] == 0
            next_rec = np.full(n, -1, np.int64)                       # record of level 1 of each failed episode
            next_rec[failed] = n + base1
        t_plan = (time.perf_counter() - t0) / n

        def adopt(which, r):                                          # planner state after a plan() call
            nps = rec["new_prev_s"][r]
            e = sel[which]
            self.prev_s[e] = np.where(np.isnan(nps), self.prev_s[e], nps)
            self.last_stats[e] = np.where(rec["stats_valid"][r][:, None] != 0, rec["stats"][r], -1)
            ok = found_all[r]
            self.last_kappa[e[ok]] = rec["new_last_kappa"][r[ok]]
            path_rec[which[ok]] = r[ok]

        adopt(everyone, cur)
        found = found_all[cur]
        issued = st0.copy()                                           # state of the configuration the attempt ran under
        sm.update(sel, found, clearance, clearance_ahead, speed)
        retries = np.zeros(n, np.int64)
        active = ~found & (sm.state[sel] != issued) & (retries < self.MAX_REPLAN) & (retries + 1 < n_lvl)
        while active.any():
            w = np.flatnonzero(active)
            cur[w] = np.where(retries[w] == 0, next_rec[w], cur[w] + 1)
            retries[w] += 1
            adopt(w, cur[w])
            ok = found_all[cur[w]]
            found[w[ok]] = True
            issued[w] = sm.state[sel[w]]
            again = w[~ok]
            if len(again):
                sm.update(sel[again], np.zeros(len(again), bool), clearance[again], clearance_ahead[again], speed[again])
            active = np.zeros(n, bool)
            active[again] = (sm.state[sel[again]] != issued[again]) & (retries[again] < self.MAX_REPLAN) & \
                            (retries[again] + 1 < n_lvl[again])
        # --- 4. ego update (:655-676) or emergency stop (:749-802)
        old_a = self.ego[sel, 4].copy()
        keep = np.where(path_rec >= 0, rec["n_keep"][np.maximum(path_rec, 0)], 0)
        follow = keep >= 2
        new_ego = self.ego[sel].copy()
        jerk = np.zeros(n)
        if follow.any():
            r = path_rec[follow]
            for col, f in enumerate(("x", "y", "yaw", "v", "a")):
                new_ego[follow, col] = rec[f][r, 1]
            jerk[follow] = (new_ego[follow, 4] - old_a[follow]) / self.dt
        brake = ~follow
        if brake.any():
            x, y, yaw, v = (self.ego[sel, k][brake] for k in range(4))
            nx, ny, nv, na = emergency_stop(x, y, yaw, v, self.last_clearance[sel][brake], c.dt, c.ego_max_accel,
                                            getattr(c, "ego_emergency_decel", None))
            new_ego[brake, 0], new_ego[brake, 1] = nx, ny
            new_ego[brake, 3], new_ego[brake, 4] = nv, na
            jerk[brake] = (na - old_a[brake]) / c.dt
            self.last_kappa[sel[brake]] = 0.0                         # planner.reset_ego_curvature()
        self.ego[sel], self.jerk[sel] = new_ego, jerk
        # --- 5. result metrics on the new ego state, goal test (:864-883)
        pending = observe(new_ego)                                    # enqueued; collected below, behind the bookkeeping
        chosen = np.maximum(path_rec, 0)
        kmax = int(keep.max()) if n else 0
        slot = np.full(len(self.episodes), -1, np.int64)
        slot[sel] = everyone
        if hasattr(self.engine, "gather_paths"):                      # one dense block, copied by the library
            block = self.engine.gather_paths(rec, chosen, kmax, out=self._history_block(len(self._steps), n, kmax))
            paths = {f: block[j] for j, f in enumerate(_abi.PATH_FIELDS)}
        else:
            paths = {f: rec[f][chosen, :kmax].copy() for f in _abi.PATH_FIELDS}
        after, s_now = pending()
        self.goal_prev_s[sel] = s_now
        self._steps.append(dict(
            time=self.time, slot=slot, off=off, ego=new_ego, jerk=jerk, state=sm.state[sel].copy(), pos=pos, vel=vel,
            pred=pred, pred_src=pred_src, after=after, stats=self.last_stats[sel].copy(), has_path=path_rec >= 0, keep=keep,
            cost=rec["cost"][chosen], paths=paths,
            t_pred=t_pred, t_plan=t_plan, sel=sel))
        collided = after["collision"] != 0
        at_goal = self.s_end - s_now < 2.0
        self.step_counts[sel] += 1
        self.termination[sel[at_goal & ~collided]] = 2
        self.termination[sel[collided]] = 1
        self.alive[sel[collided | at_goal]] = False
        self.time += self.dt
        return n

    def _record(self, k: int, e: int) -> StepRecord:
        """StepRecord of episode e at lock step k, from the step's arrays."""
        s = self._steps[k]
        i = int(s["slot"][e])
        lo, hi = int(s["off"][i]), int(s["off"][i + 1])
        ego = EgoVehicleState(*(float(v) for v in s["ego"][i]), jerk=float(s["jerk"][i]), timestamp=s["time"] + self.dt)
        ego.state = _STATES[int(s["state"][i])]
        path = None
        if s["has_path"][i]:
            kn = int(s["keep"][i])
            path = FrenetPath(**{f: s["paths"][f][i, :kn].copy() for f in _abi.PATH_FIELDS})
            path.cost = float(s["cost"][i])
        a = s["after"][i]
        m = {"min_distance": float(a["min_distance"]), "collision": bool(a["collision"]), "ttc": float(a["ttc"]),
             "clearance": float(a["clearance"]), "clearance_ahead": float(a["clearance_ahead"])}
        if s["stats"][i, 0] >= 0:
            m["n_collision_rejected"] = int(s["stats"][i, _abi.ST_COLLISION])
        p = self.peds[e]
        if s["pred"] is None and s.get("pred_src") is not None:       # fused step: the prediction stayed in HBM
            s["pred"] = self._materialise_prediction(s["pred_src"], s["off"])
            s["pred_src"] = None
        return StepRecord(s["time"], ego, s["pos"][lo:hi].copy(), s["vel"][lo:hi].copy(), p.goals.copy(),
                          None if s["pred"] is None else s["pred"][lo:hi], path, m,
                          {"prediction": s["t_pred"], "planning": s["t_plan"]})

    def run(self, n_steps: Optional[int] = None) -> List[EpisodeHistory]:
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
        history = list(history)                                       # (a lazy EpisodeHistory builds its records once)

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

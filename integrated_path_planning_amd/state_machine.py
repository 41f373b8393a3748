"""Fail-safe state machine and the escalate-and-retry planning cycle.

``FailSafeStateMachine`` restates the reference's scalar control logic (src/core/state_machine.py:29-278):
NORMAL / CAUTION / EMERGENCY, the planner configuration each state issues, the transitions on plan success,
failure and clearance.  ``SpeculativePlanningCycle`` restates ``IntegratedSimulator._execute_planning_cycle``
(integrated_simulator.py:529-653) the MI355X way: the configurations of every escalation level that the loop
could reach are known before planning (they depend only on the state and on the current safety metrics), so
all of them are planned in ONE launch -- same ego, same obstacles, chained nearest-point cache -- and the
reference's control flow is then replayed on the results.  A failing step costs one launch instead of up to four
sequential ``plan()`` calls.
"""
from __future__ import annotations

import copy
import math
from dataclasses import dataclass
from enum import Enum, auto
from typing import Any, Dict, List, Optional, Tuple

import numpy as np

from .batch import PlanRequest
from .data_structures import FrenetPath


class VehicleState(Enum):
    NORMAL = auto()
    CAUTION = auto()
    EMERGENCY = auto()


@dataclass
class StateMachineOutput:
    state: VehicleState
    target_speed_override: Optional[float] = None
    constraint_overrides: Optional[Dict[str, float]] = None
    max_stop_distance: Optional[float] = None


class FailSafeStateMachine:
    """Same constructor contract as the reference: a config object read with getattr (state_machine.py:32-98)."""

    def __init__(self, config) -> None:
        self.config = config
        self.current_state = VehicleState.NORMAL
        self.consecutive_failures = 0
        fp_mode = getattr(config, "ego_footprint", None)
        if fp_mode is not None and fp_mode != "circle":
            seg = config.vehicle_length / config.ego_footprint_n_circles          # footprint.py:37-39
            ego_radius = float(np.hypot(seg / 2, config.vehicle_width / 2))
        else:
            ego_radius = getattr(config, "ego_radius", 1.0)
        combined = ego_radius + getattr(config, "ped_radius", 0.2)
        rc = getattr(config, "state_machine_recover_clearance_caution", None)
        re = getattr(config, "state_machine_recover_clearance_emergency", None)
        self.clearance_caution = rc if rc is not None else getattr(config, "state_machine_safe_distance_caution", 2.0) - combined
        self.clearance_emergency = re if re is not None else getattr(config, "state_machine_safe_distance_emergency", 3.0) - combined
        self.trigger_clearance_caution = getattr(config, "state_machine_trigger_clearance_caution", 0.0)
        self.trigger_time_headway = getattr(config, "state_machine_trigger_time_headway", 0.0)
        self.envelope_decel = getattr(config, "state_machine_envelope_decel", 0.0)
        self.envelope_standoff = getattr(config, "state_machine_envelope_standoff", 0.5)
        self._last_clearance = float("inf")
        self._last_clearance_ahead = float("inf")

    def observe_metrics(self, safety_metrics: Dict[str, Any]) -> None:                # :99-113
        self._last_clearance = safety_metrics.get("clearance", float("inf"))
        self._last_clearance_ahead = safety_metrics.get("clearance_ahead", self._last_clearance)

    def update(self, plan_found: bool, safety_metrics: Dict[str, Any], ego_speed: float = 0.0) -> StateMachineOutput:
        """Transitions of state_machine.py:116-179."""
        self.observe_metrics(safety_metrics)
        trigger = self.trigger_clearance_caution + self.trigger_time_headway * max(ego_speed, 0.0)
        clearance = safety_metrics.get("clearance", float("inf"))
        if self.current_state == VehicleState.NORMAL:
            if not plan_found:
                self.current_state = VehicleState.CAUTION
                self.consecutive_failures += 1
            elif trigger > 0.0 and clearance < trigger:
                self.current_state = VehicleState.CAUTION                             # preventive escalation
                self.consecutive_failures = 0
            else:
                self.consecutive_failures = 0
        elif self.current_state == VehicleState.CAUTION:
            if plan_found and self.consecutive_failures == 0:
                if clearance > max(self.clearance_caution, trigger):
                    self.current_state = VehicleState.NORMAL
            elif not plan_found:
                self.current_state = VehicleState.EMERGENCY
                self.consecutive_failures += 1
            else:
                self.consecutive_failures = 0
        elif self.current_state == VehicleState.EMERGENCY:
            if plan_found and clearance > self.clearance_emergency:
                self.current_state = VehicleState.CAUTION
        return self._get_planner_config()

    def _envelope_speed(self) -> Optional[float]:                                     # :249-264
        if self.envelope_decel <= 0.0 or not math.isfinite(self._last_clearance_ahead):
            return None
        room = max(self._last_clearance_ahead - self.envelope_standoff, 0.0)
        return math.sqrt(2.0 * self.envelope_decel * room)

    def _stop_room_to_pedestrian(self) -> Optional[float]:                            # :266-278
        if not math.isfinite(self._last_clearance_ahead):
            return None
        return max(self._last_clearance_ahead - 0.2, 0.05)

    def _get_planner_config(self) -> StateMachineOutput:                              # :181-247
        cfg = self.config
        if self.current_state == VehicleState.NORMAL:
            override = None
            v_env = self._envelope_speed()
            if v_env is not None and v_env < cfg.ego_target_speed:
                override = v_env
            return StateMachineOutput(VehicleState.NORMAL, override, None)
        if self.current_state == VehicleState.CAUTION:
            accel_mult = getattr(cfg, "state_machine_caution_accel_multiplier", 1.5)
            speed_mult = getattr(cfg, "state_machine_caution_speed_multiplier", 0.8)
            target = cfg.ego_target_speed * speed_mult
            stop = None
            v_env = self._envelope_speed()
            if v_env is not None:
                target = min(target, v_env)
                if v_env <= 0.0:
                    stop = self._stop_room_to_pedestrian()
            return StateMachineOutput(VehicleState.CAUTION, target,
                                      {"max_accel": cfg.ego_max_accel * accel_mult,
                                       "max_speed": cfg.ego_max_speed * speed_mult}, stop)
        accel_mult = getattr(cfg, "state_machine_emergency_accel_multiplier", 3.0)
        lat_mult = getattr(cfg, "state_machine_emergency_lat_accel_multiplier", 2.0)
        return StateMachineOutput(VehicleState.EMERGENCY, 0.0,
                                  {"max_accel": cfg.ego_max_accel * accel_mult,
                                   "max_lat_accel": getattr(cfg, "ego_max_lat_accel", 3.0) * lat_mult},
                                  self._stop_room_to_pedestrian() if self.envelope_decel > 0.0 else None)


@dataclass
class CycleResult:
    planned_path: Optional[FrenetPath]
    attempts: int                         # plan() calls the reference would have made this step
    retries: int                          # of which escalation retries
    states: List[VehicleState]            # state each attempt was planned in
    final_output: StateMachineOutput      # what the state machine issues after the step


class SpeculativePlanningCycle:
    """One step of _execute_planning_cycle with every reachable escalation level planned in a single launch."""

    def __init__(self, planner, state_machine: FailSafeStateMachine, ego_target_speed: float,
                 max_replan_attempts: int = 3):
        self.planner = planner                    # integrated_path_planning_amd.FrenetPlanner
        self.sm = state_machine
        self.ego_target_speed = ego_target_speed
        self.max_replan_attempts = max_replan_attempts
        self._path_kw = {}                        # BatchedClosedLoop asks for NumPy rows instead of lists

    def _ladder(self, metrics, ego_speed) -> List[StateMachineOutput]:
        """Configurations the retry loop would issue if every attempt failed (dry run on a copy)."""
        sm = copy.copy(self.sm)
        out = [sm._get_planner_config()]
        retries = 0
        cur = out[0]
        nxt = sm.update(False, metrics, ego_speed)
        while nxt.state != cur.state and retries < self.max_replan_attempts:
            out.append(nxt)
            retries += 1
            cur = nxt
            nxt = sm.update(False, metrics, ego_speed)
        return out

    def prepare(self, ego_state, static_obstacles, dynamic_obstacles, current_metrics: Dict[str, Any],
                dynamic_obstacles_distribution=None, replan_attempts_used: int = 0):
        """The requests of every escalation level this step can reach (first = the current state's), and the ladder
        they were built from.  Requests after the first chain their nearest-point cache on the one before."""
        pl = self.planner
        budget = max(self.max_replan_attempts - replan_attempts_used, 0)
        ladder = self._ladder(current_metrics, ego_state.v)[: 1 + budget]
        reqs = []
        for j, cfg in enumerate(ladder):
            target = cfg.target_speed_override if cfg.target_speed_override is not None else self.ego_target_speed
            reqs.append(PlanRequest(
                x=float(ego_state.x), y=float(ego_state.y), yaw=float(ego_state.yaw), v=float(ego_state.v),
                a=float(ego_state.a), target_speed=float(target), last_kappa=float(pl._last_kappa),
                prev_s=None if j else getattr(pl.converter, "_prev_s", None), chain_prev_s=bool(j),
                overrides=cfg.constraint_overrides, max_stop_distance=cfg.max_stop_distance,
                static=static_obstacles, dyn=dynamic_obstacles, dist=dynamic_obstacles_distribution))
        return ladder, reqs, budget

    def finish(self, ladder, budget, res, base: int, ego_state, current_metrics: Dict[str, Any]) -> CycleResult:
        """Replay of integrated_simulator.py:576-653 on the speculative results res.records[base + j]."""
        pl = self.planner

        def adopt(j):
            rec = res.records[base + j]
            if not np.isnan(rec.new_prev_s):
                pl.converter._prev_s = float(rec.new_prev_s)
            pl.last_check_stats = res.stats(base + j)
            path = res.path(base + j, **self._path_kw)
            if path is not None:
                pl._last_kappa = float(rec.new_last_kappa)
            return path

        states = [ladder[0].state]
        path = adopt(0)
        sm_output = ladder[0]
        new_output = self.sm.update(path is not None, current_metrics, ego_speed=ego_state.v)
        retries, j = 0, 0
        while path is None and new_output.state != sm_output.state and retries < budget:
            j += 1
            retries += 1
            states.append(new_output.state)
            path = adopt(j)
            if path is not None:
                break
            sm_output = new_output
            new_output = self.sm.update(False, current_metrics, ego_speed=ego_state.v)
        return CycleResult(path, 1 + retries, retries, states, new_output)

    def execute(self, ego_state, static_obstacles, dynamic_obstacles, current_metrics: Dict[str, Any],
                dynamic_obstacles_distribution=None, replan_attempts_used: int = 0) -> CycleResult:
        ladder, reqs, budget = self.prepare(ego_state, static_obstacles, dynamic_obstacles, current_metrics,
                                            dynamic_obstacles_distribution, replan_attempts_used)
        res = self.planner.engine.plan_batch(reqs)
        return self.finish(ladder, budget, res, 0, ego_state, current_metrics)

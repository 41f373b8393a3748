"""Carrier types of the planner boundary.

Same field names and behaviour as the reference's planner-facing types
(src/core/data_structures.py:32-62 EgoVehicleState, :119-146 FrenetState,
:149-220 FrenetPath) so that callers written against the reference
(IntegratedSimulator._update_ego_state, the animator) work unchanged.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np


@dataclass
class EgoVehicleState:
    x: float
    y: float
    yaw: float
    v: float
    a: float
    jerk: float = 0.0
    timestamp: float = 0.0
    state: object = None

    def to_array(self) -> np.ndarray:
        return np.array([self.x, self.y, self.yaw, self.v, self.a, self.jerk])

    @classmethod
    def from_array(cls, arr, timestamp: float = 0.0) -> "EgoVehicleState":
        jerk = arr[5] if len(arr) > 5 else 0.0
        return cls(x=arr[0], y=arr[1], yaw=arr[2], v=arr[3], a=arr[4], jerk=jerk, timestamp=timestamp)


@dataclass
class FrenetState:
    s: float
    s_d: float
    s_dd: float
    d: float
    d_d: float
    d_dd: float

    def to_array(self) -> np.ndarray:
        return np.array([self.s, self.s_d, self.s_dd, self.d, self.d_d, self.d_dd])

    @classmethod
    def from_array(cls, arr) -> "FrenetState":
        return cls(*[float(v) for v in arr[:6]])


def _len(seq) -> int:
    try:
        return len(seq)
    except TypeError:
        return 0


@dataclass
class PedestrianState:
    """What a pedestrian source's ``get_state()`` returns (reference: data_structures.py:66-109): positions,
    velocities, goals [n_peds, 2], ids [n_peds] (0..n-1 when not given), timestamp."""
    positions: np.ndarray
    velocities: np.ndarray
    goals: np.ndarray
    ids: Optional[np.ndarray] = None
    timestamp: float = 0.0

    def __post_init__(self):
        for name in ("positions", "velocities", "goals"):
            arr = getattr(self, name)
            if arr.ndim != 2 or arr.shape[1] != 2:
                raise AssertionError(f"{name} must be (n_peds, 2)")
        if not (len(self.positions) == len(self.velocities) == len(self.goals)):
            raise AssertionError("All arrays must have same number of pedestrians")
        if self.ids is None:
            self.ids = np.arange(self.n_peds)

    @property
    def n_peds(self) -> int:
        return int(self.positions.shape[0])

    @property
    def pedestrians(self) -> np.ndarray:
        return self.positions

    def to_social_force_format(self) -> np.ndarray:
        return np.hstack([self.positions, self.velocities, self.goals])


@dataclass
class FrenetPath:
    t: List[float] = field(default_factory=list)
    s: List[float] = field(default_factory=list)
    s_d: List[float] = field(default_factory=list)
    s_dd: List[float] = field(default_factory=list)
    s_ddd: List[float] = field(default_factory=list)
    d: List[float] = field(default_factory=list)
    d_d: List[float] = field(default_factory=list)
    d_dd: List[float] = field(default_factory=list)
    d_ddd: List[float] = field(default_factory=list)
    x: List[float] = field(default_factory=list)
    y: List[float] = field(default_factory=list)
    yaw: List[float] = field(default_factory=list)
    v: List[float] = field(default_factory=list)
    a: List[float] = field(default_factory=list)
    c: List[float] = field(default_factory=list)
    cost: float = float("inf")

    def __len__(self) -> int:
        if _len(self.t) == 0:
            return 0
        return min(_len(self.t), _len(self.x), _len(self.y), _len(self.yaw), _len(self.v), _len(self.a))

    def get_state_at_index(self, idx: int) -> EgoVehicleState:
        if idx < 0 or idx >= len(self):
            raise IndexError(f"Index {idx} out of range for path of length {len(self)}")
        return EgoVehicleState(x=self.x[idx], y=self.y[idx], yaw=self.yaw[idx], v=self.v[idx], a=self.a[idx],
                               timestamp=self.t[idx])

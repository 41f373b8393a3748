"""Safety metrics behind the reference's call surface.

``compute_safety_metrics_static`` has the signature and result dictionary of
src/core/data_structures.py:301-388; the arithmetic runs in ``k_safety``
(csrc/fot_kernels.hip), one wavefront per ego, through ``fot_safety_metrics_batch``.
``SafetyMonitor.metrics_batch`` is the many-ego form the reference lacks.
"""
from __future__ import annotations

from typing import Any, Dict, Optional, Sequence

import numpy as np

from .footprint import EgoFootprint
from .planner import BatchPlanner

_STRAIGHT = (np.array([0.0, 1.0]), np.array([0.0, 0.0]))      # a handle needs a reference path; k_safety never reads it


class SafetyMonitor:
    """One libfot handle per footprint (the footprint is a handle constant, include/fot.h fot_params)."""

    def __init__(self, footprint: Optional[EgoFootprint] = None, engine: Optional[BatchPlanner] = None, device: int = -1):
        self.footprint = footprint
        if engine is None:
            kw = {"footprint": footprint} if footprint is not None else {}
            engine = BatchPlanner(waypoints=_STRAIGHT, device=device, **kw)
        self.engine = engine

    def metrics_batch(self, egos, ped_positions: Sequence, ped_velocities: Sequence, ego_radius: float,
                      ped_radius: float) -> np.ndarray:
        return self.engine.safety_metrics(egos, ped_positions, ped_velocities, ego_radius, ped_radius,
                                          use_footprint=self.footprint is not None)

    def metrics(self, ego_state, ped_state, ego_radius: float, ped_radius: float) -> Dict[str, Any]:
        r = self.metrics_batch([[ego_state.x, ego_state.y, ego_state.yaw, ego_state.v]], [ped_state.positions],
                               [ped_state.velocities], ego_radius, ped_radius)[0]
        return {"min_distance": float(r["min_distance"]), "collision": bool(r["collision"]), "ttc": float(r["ttc"]),
                "clearance": float(r["clearance"]), "clearance_ahead": float(r["clearance_ahead"])}


_monitors: Dict[Any, SafetyMonitor] = {}


def compute_safety_metrics_static(ego_state, ped_state, ego_radius: float, ped_radius: float,
                                  footprint: Optional[EgoFootprint] = None) -> Dict[str, Any]:
    """Drop-in for data_structures.py:301 (same arguments, same keys)."""
    key = None if footprint is None else (float(footprint.radius), tuple(float(o) for o in footprint.offsets))
    mon = _monitors.get(key)
    if mon is None:
        mon = _monitors[key] = SafetyMonitor(footprint)
    return mon.metrics(ego_state, ped_state, ego_radius, ped_radius)

"""Multi-circle ego footprint (reference: src/core/footprint.py:14-45)."""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass(frozen=True)
class EgoFootprint:
    offsets: np.ndarray     # circle centres along the heading axis, relative to the vehicle centre [m]
    radius: float           # common circle radius [m]

    @classmethod
    def multi_circle(cls, vehicle_length: float, vehicle_width: float, n_circles: int) -> "EgoFootprint":
        """n equal circles, each circumscribing one L/n x W slice of the rectangle."""
        if n_circles < 1:
            raise ValueError(f"n_circles must be >= 1, got {n_circles}")
        seg = vehicle_length / n_circles
        offsets = -vehicle_length / 2 + seg / 2 + seg * np.arange(n_circles)
        return cls(offsets=offsets, radius=float(np.hypot(seg / 2, vehicle_width / 2)))

    def circle_centers(self, x: float, y: float, yaw: float) -> np.ndarray:
        direction = np.array([np.cos(yaw), np.sin(yaw)])
        return np.array([x, y]) + self.offsets[:, None] * direction

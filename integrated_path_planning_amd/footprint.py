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
        half_slice = 0.5 * vehicle_length / n_circles             # half the length of one slice
        # centre k sits in the middle of slice k, counted from the rear end at -L/2: -L/2 + (2k + 1) half_slice
        k = np.arange(n_circles)
        return cls(offsets=(2 * k + 1) * half_slice - 0.5 * vehicle_length,
                   radius=float(np.sqrt(half_slice ** 2 + (0.5 * vehicle_width) ** 2)))

    def circle_centers(self, x: float, y: float, yaw: float) -> np.ndarray:
        """[n_circles, 2]: the centres of a vehicle at (x, y) heading yaw."""
        out = np.empty((len(self.offsets), 2))
        out[:, 0] = x + self.offsets * np.cos(yaw)
        out[:, 1] = y + self.offsets * np.sin(yaw)
        return out

"""Synthetic workloads for the planner hot path.

Shared by bench.py, the parity tests and tests/golden/make_golden.py so that
every leg (reference, oracle, HIP) sees bit-identical inputs.  Pure NumPy.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import numpy as np

# straight reference path of configs 2-5: knots every 10 m along y = 0
STRAIGHT_WX = np.arange(0.0, 101.0, 10.0)
STRAIGHT_WY = np.zeros_like(STRAIGHT_WX)
# curved variant (scenario_03 waypoints), parity only
CURVED_WX = np.array([-30.0, -8.0, -0.4, 2.0, 2.0])
CURVED_WY = np.array([2.5, 2.5, -0.4, -8.0, -30.0])

TARGET_SPEED = 30.0 / 3.6

# planner keyword sets
CONFIG2_PLANNER = dict(dt=0.1)                                       # module defaults otherwise
CONFIG3_PLANNER = dict(dt=0.1, robot_radius=1.0, obstacle_radius=0.2, chance_epsilon=0.0)


@dataclass
class Instance:
    """One ego/scenario instance: ego state + its obstacle tensors."""
    ego: np.ndarray                         # x, y, yaw, v, a
    static: np.ndarray = field(default_factory=lambda: np.empty((0, 2)))
    dyn: Optional[np.ndarray] = None        # [P, T, 2]
    dist: Optional[np.ndarray] = None       # [S, P, T, 2]
    target_speed: float = TARGET_SPEED


def _ego(rng) -> np.ndarray:
    return np.array([rng.uniform(0.0, 40.0), rng.uniform(-1.0, 1.0), rng.uniform(-0.1, 0.1),
                     rng.uniform(2.0, 8.0), rng.uniform(-1.0, 1.0)])


def config2_instance(seed: int, n_static: int = 10) -> Instance:
    """1 ego + 10 static points (BASELINE config 2)."""
    rng = np.random.default_rng(seed)
    ego = _ego(rng)
    sx = rng.uniform(ego[0] + 5.0, ego[0] + 45.0, n_static)
    sy = rng.uniform(-8.0, 8.0, n_static)
    return Instance(ego=ego, static=np.stack([sx, sy], axis=1))


def config3_instance(seed: int, S: int = 20, P: int = 30, T: int = 51, dt: float = 0.1,
                     dtype=np.float32) -> Instance:
    """1 ego + S-sample prediction distribution of P pedestrians over T steps
    (BASELINE configs 3-5).  Values are generated in float64 and rounded to
    ``dtype``; every consumer sees the rounded values."""
    rng = np.random.default_rng(seed)
    ego = _ego(rng)
    p0x = rng.uniform(ego[0] - 10.0, ego[0] + 90.0, P)
    p0y = rng.uniform(-25.0, 25.0, P)
    speed = rng.normal(1.3, 0.2, P)
    heading = rng.uniform(0.0, 2.0 * np.pi, P)
    dv = rng.normal(0.0, 0.3, (S, P, 1, 2))
    walk = rng.normal(0.0, 0.05, (S, P, T, 2))
    walk[:, :, 0, :] = 0.0
    walk = np.cumsum(walk, axis=2)
    vel = np.stack([speed * np.cos(heading), speed * np.sin(heading)], axis=1)        # [P, 2]
    t = (np.arange(T) * dt)[None, None, :, None]
    p0 = np.stack([p0x, p0y], axis=1)[None, :, None, :]
    dist = p0 + (vel[None, :, None, :] + dv) * t + walk
    dist = dist.astype(dtype)
    return Instance(ego=ego, dist=dist, dyn=dist[0])


def config3_batch(seeds, **kw):
    return [config3_instance(int(s), **kw) for s in seeds]


def lattice_size(target_speed=TARGET_SPEED, dt=0.1, min_t=4.0, max_t=5.0, d_t_s=5.0 / 3.6,
                 d_road_w=0.5, max_road_width=7.0, moving=True) -> int:
    """Candidates per plan for the given sampling (frenet_planner.py:397-420, 469-475)."""
    n_ti = int((max_t - min_t) / dt + 1e-9) + 1
    n_down = int(target_speed / d_t_s + 1e-9)
    n_tv = n_down + 1 + (1 if target_speed - n_down * d_t_s > 1e-9 else 0)
    n_di = 2 * int(max_road_width / d_road_w + 1e-9) + 1
    n_brake = len(np.arange(0.5, min_t - 1e-9, 0.5)) if moving else 0
    return n_ti * n_tv * n_di + n_brake

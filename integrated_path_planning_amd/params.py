"""Planner parameters -> ``fot_params`` (defaults = the reference's module constants,
frenet_planner.py:25-46, 91)."""
from __future__ import annotations

from typing import Optional, Sequence

from . import _abi

MAX_SPEED = 50.0 / 3.6
MAX_ACCEL = 2.0
MAX_CURVATURE = 1.0
MAX_ROAD_WIDTH = 7.0
D_ROAD_W = 0.5
DT = 0.2
MAX_T = 5.0
MIN_T = 4.0
TARGET_SPEED = 30.0 / 3.6
D_T_S = 5.0 / 3.6
N_S_SAMPLE = 1
K_J = 0.1
K_T = 0.1
K_D = 1.0
K_S_DOT = 1.0
K_LAT = 1.0
K_LON = 1.0
ROBOT_RADIUS = 2.0
MAX_LAT_ACCEL = 3.0


def make_params(max_speed=MAX_SPEED, max_accel=MAX_ACCEL, max_curvature=MAX_CURVATURE, dt=DT,
                d_road_w=D_ROAD_W, max_road_width=MAX_ROAD_WIDTH, robot_radius=ROBOT_RADIUS,
                obstacle_radius=0.3, min_t=MIN_T, max_t=MAX_T, d_t_s=D_T_S, n_s_sample=N_S_SAMPLE,
                max_lat_accel=MAX_LAT_ACCEL, k_j=K_J, k_t=K_T, k_d=K_D, k_s_dot=K_S_DOT, k_lat=K_LAT,
                k_lon=K_LON, chance_epsilon=0.0, collision_margin_inflation=1.0,
                footprint_offsets: Optional[Sequence[float]] = None, footprint_radius: float = 0.0) -> _abi.Params:
    p = _abi.Params()
    p.max_speed, p.max_accel, p.max_curvature = float(max_speed), float(max_accel), float(max_curvature)
    p.max_lat_accel = float(max_lat_accel)
    p.dt, p.d_road_w, p.max_road_width = float(dt), float(d_road_w), float(max_road_width)
    p.robot_radius, p.obstacle_radius = float(robot_radius), float(obstacle_radius)
    p.min_t, p.max_t, p.d_t_s = float(min_t), float(max_t), float(d_t_s)
    p.k_j, p.k_t, p.k_d, p.k_s_dot = float(k_j), float(k_t), float(k_d), float(k_s_dot)
    p.k_lat, p.k_lon = float(k_lat), float(k_lon)
    p.chance_epsilon = float(chance_epsilon)
    p.collision_margin_inflation = float(collision_margin_inflation)
    if footprint_offsets is not None:
        offs = [float(o) for o in footprint_offsets]
        if not 1 <= len(offs) <= _abi.MAX_CIRCLES:
            raise ValueError(f"footprint needs 1..{_abi.MAX_CIRCLES} circles, got {len(offs)}")
        p.n_circles = len(offs)
        p.footprint_radius = float(footprint_radius)
        for i, o in enumerate(offs):
            p.footprint_offsets[i] = o
    return p

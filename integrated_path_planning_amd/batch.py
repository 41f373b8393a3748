"""Packing of independent ego/scenario instances into one ``fot_batch`` (include/fot.h).

One instance = the arguments of one ``FrenetPlanner.plan()`` call of the
reference (frenet_planner.py:227-236).  Obstacle tensors of all instances are
concatenated; shapes travel as host metadata.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from . import _abi


@dataclass
class PlanRequest:
    """Arguments of one plan() call plus the planner's cross-call state."""
    x: float
    y: float
    yaw: float
    v: float
    a: float
    target_speed: float = 30.0 / 3.6
    last_kappa: float = 0.0
    prev_s: Optional[float] = None
    chain_prev_s: bool = False                   # prev_s := new_prev_s of the previous request (next call on the same planner)
    is_frenet: bool = False                      # x, y, yaw, v, a, last_kappa hold s, s_d, s_dd, d, d_d, d_dd (FOT_EGO_IS_FRENET)
    overrides: Optional[dict] = None
    max_stop_distance: Optional[float] = None
    static: Optional[np.ndarray] = None          # [Ns, 2]
    dyn: Optional[np.ndarray] = None             # [P, T, 2]
    dist: Optional[np.ndarray] = None            # [S, P, T, 2]


def _dyn_of(req: PlanRequest):
    """Which dynamic tensor plan() would use (frenet_planner.py:1043-1047, 1205-1208).  Passed on as it is: the
    reference's rule for NaN coordinates (a pedestrian whose track holds one is ignored at every time step,
    frenet_planner.py:1211-1219) is applied by the library itself, for every producer of the tensor."""
    if req.dist is not None and np.size(req.dist) > 0:
        d = np.asarray(req.dist)
        if d.ndim != 4 or d.shape[-1] != 2:
            raise ValueError(f"distribution must be [S, P, T, 2], got {d.shape}")
        return _abi.DYN_DISTRIBUTION, d
    if req.dyn is not None and np.size(req.dyn) > 0 and np.shape(req.dyn)[-1] == 2:
        d = np.asarray(req.dyn)
        if d.ndim != 3:
            raise ValueError(f"dynamic obstacles must be [P, T, 2], got {d.shape}")
        return _abi.DYN_SINGLE, d[None]
    return _abi.DYN_NONE, None


class PackedBatch:
    """Host-side ``fot_batch`` with the NumPy buffers that back its pointers."""

    def __init__(self, requests: Sequence[PlanRequest], obstacle_dtype=np.float64, dyn_layout_tsp: bool = False):
        """dyn_layout_tsp: pack every dynamic tensor time-major, [T][S][P][2] (``FOT_DYN_LAYOUT_TSP``) -- the layout the
        broad phase reads with fully coalesced loads and the device-side resampler can write directly."""
        n = len(requests)
        self.dyn_layout_tsp = bool(dyn_layout_tsp)
        self.n = n
        self.np_dtype = np.dtype(obstacle_dtype)
        if self.np_dtype not in (np.dtype(np.float32), np.dtype(np.float64)):
            raise ValueError("obstacle_dtype must be float32 or float64")
        self.ego = (_abi.Ego * max(n, 1))()
        self.target = np.zeros(max(n, 1), dtype=np.float64)
        self.overrides = (_abi.Overrides * max(n, 1))()
        self.max_stop = np.full(max(n, 1), np.nan, dtype=np.float64)
        self.static_off = np.zeros(n + 1, dtype=np.int32)
        self.dyn_off = np.zeros(max(n, 1), dtype=np.int64)
        self.dyn_dims = np.zeros((max(n, 1), 4), dtype=np.int32)
        statics: List[np.ndarray] = []
        dyns: List[np.ndarray] = []
        dyn_cursor = 0
        nan = float("nan")
        shared = {}                                  # (id(array), mode) -> offset: requests passing the SAME dynamic tensor
                                                     # object (escalation retries of one step) share one copy of it
        for i, r in enumerate(requests):
            e = self.ego[i]
            e.x, e.y, e.yaw, e.v, e.a = float(r.x), float(r.y), float(r.yaw), float(r.v), float(r.a)
            e.last_kappa = float(r.last_kappa)
            if r.is_frenet:
                e.has_prev_s, e.prev_s = _abi.EGO_IS_FRENET, 0.0
            elif r.chain_prev_s:
                if i == 0:
                    raise ValueError("the first request of a batch cannot chain its nearest-point cache")
                e.has_prev_s, e.prev_s = 2, 0.0
            else:
                e.has_prev_s = 0 if r.prev_s is None else 1
                e.prev_s = 0.0 if r.prev_s is None else float(r.prev_s)
            self.target[i] = float(r.target_speed)
            ov = r.overrides or {}
            o = self.overrides[i]
            o.max_speed = float(ov.get("max_speed", nan))
            o.max_accel = float(ov.get("max_accel", nan))
            o.max_curvature = float(ov.get("max_curvature", nan))
            o.max_lat_accel = float(ov.get("max_lat_accel", nan))
            if r.max_stop_distance is not None:
                self.max_stop[i] = float(r.max_stop_distance)
            st = np.empty((0, 2)) if r.static is None or len(r.static) == 0 else np.asarray(r.static).reshape(-1, 2)
            statics.append(st.astype(self.np_dtype, copy=False))
            self.static_off[i + 1] = self.static_off[i] + st.shape[0]
            mode, d = _dyn_of(r)
            self.dyn_off[i] = dyn_cursor
            if mode != _abi.DYN_NONE:
                S, P, T = d.shape[0], d.shape[1], d.shape[2]
                self.dyn_dims[i] = (mode | (_abi.DYN_LAYOUT_TSP if self.dyn_layout_tsp else 0), S, P, T)
                src = r.dist if mode == _abi.DYN_DISTRIBUTION else r.dyn
                key = (id(src), mode)
                if key in shared:
                    self.dyn_off[i] = shared[key]
                else:
                    shared[key] = dyn_cursor
                    if self.dyn_layout_tsp:
                        d = np.transpose(d, (2, 0, 1, 3))                  # [S, P, T, 2] -> [T, S, P, 2]
                    dyns.append(np.ascontiguousarray(d, dtype=self.np_dtype).reshape(-1, 2))
                    dyn_cursor += S * P * T
        self.static_xy = (np.concatenate(statics, axis=0) if statics else np.empty((0, 2))).astype(self.np_dtype)
        self.static_xy = np.ascontiguousarray(self.static_xy)
        self.dyn_xy = np.ascontiguousarray(np.concatenate(dyns, axis=0) if dyns else np.empty((0, 2), self.np_dtype))
        self.n_candidates_hint = None
        self.c = self._make_struct(self.static_xy.ctypes.data if self.static_xy.size else None,
                                   self.dyn_xy.ctypes.data if self.dyn_xy.size else None)

    def _make_struct(self, static_ptr, dyn_ptr) -> _abi.Batch:
        b = _abi.Batch()
        b.n_inst = self.n
        b.obstacle_dtype = _abi.F32 if self.np_dtype == np.dtype(np.float32) else _abi.F64
        b.ego = C.cast(self.ego, C.POINTER(_abi.Ego))
        b.target_speed = self.target.ctypes.data_as(C.POINTER(C.c_double))
        b.overrides = C.cast(self.overrides, C.POINTER(_abi.Overrides))
        b.max_stop_distance = self.max_stop.ctypes.data_as(C.POINTER(C.c_double))
        b.static_xy = static_ptr
        b.static_off = self.static_off.ctypes.data_as(C.POINTER(C.c_int32)) if static_ptr else None
        b.dyn_xy = dyn_ptr
        b.dyn_off = self.dyn_off.ctypes.data_as(C.POINTER(C.c_int64)) if dyn_ptr else None
        b.dyn_dims = self.dyn_dims.ctypes.data_as(C.POINTER(C.c_int32)) if dyn_ptr else None
        return b

    def with_device_obstacles(self, static_dev_ptr: Optional[int], dyn_dev_ptr: Optional[int]) -> _abi.Batch:
        """Same batch with the obstacle coordinates already resident in HBM."""
        return self._make_struct(static_dev_ptr if self.static_xy.size else None,
                                 dyn_dev_ptr if self.dyn_xy.size else None)


def request_from_instance(inst, **kw) -> PlanRequest:
    """plan() arguments of a ``synthetic.Instance`` (the distribution wins over the single sample, as in the
    reference: frenet_planner.py:1043-1047)."""
    e = inst.ego
    return PlanRequest(x=e[0], y=e[1], yaw=e[2], v=e[3], a=e[4], target_speed=inst.target_speed,
                       static=inst.static, dyn=None if inst.dist is not None else inst.dyn, dist=inst.dist, **kw)

"""Host side of the MI355X Frenet planner.

``BatchPlanner`` owns one libfot handle (one GPU) and plans many independent
ego/scenario instances per call.  ``FrenetPlanner`` keeps the call surface of the
reference class (src/planning/frenet_planner.py:125-304) on top of it, so it
drops into ``IntegratedSimulator`` (integrated_simulator.py:342-366, 576-584,
622-630, 732, 802) unchanged.  All planning arithmetic runs in the gfx950
kernels; nothing here computes a trajectory.
"""
from __future__ import annotations

import ctypes as C
import sys
from typing import Dict, List, Optional, Sequence

import numpy as np

from . import _abi
from .batch import PackedBatch, PlanRequest
from .data_structures import EgoVehicleState, FrenetPath, FrenetState
from .params import (D_ROAD_W, D_T_S, DT, MAX_ACCEL, MAX_CURVATURE, MAX_ROAD_WIDTH, MAX_SPEED, MAX_T, MIN_T,
                     N_S_SAMPLE, ROBOT_RADIUS, TARGET_SPEED, make_params)

from dataclasses import dataclass

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int32)


# carrier types of the reference's polynomial builders (frenet_planner.py:95-123), for the shim's views of them
@dataclass(frozen=True)
class TimeCache:
    t: np.ndarray
    t2: np.ndarray
    t3: np.ndarray
    t4: np.ndarray
    t5: np.ndarray
    quartic_A_inv: np.ndarray
    quintic_A_inv: np.ndarray


@dataclass(frozen=True)
class LongitudinalProfile:
    t: np.ndarray
    s: np.ndarray
    s_d: np.ndarray
    s_dd: np.ndarray
    s_ddd: np.ndarray


@dataclass(frozen=True)
class LateralProfile:
    d: np.ndarray
    d_d: np.ndarray
    d_dd: np.ndarray
    d_ddd: np.ndarray


def _as_dp(a: np.ndarray):
    return a.ctypes.data_as(_dp)


def _addr(a: np.ndarray) -> int:
    """Address of a C-contiguous array.  Plain dtypes: through ``__array_interface__`` (``ndarray.ctypes`` builds a
    helper object per access); record dtypes: through ``ndarray.ctypes`` (the interface spells out the whole record
    description on every access: 10-25 us for the ego / result records, against 1-2 us)."""
    return a.ctypes.data if a.dtype.names else a.__array_interface__["data"][0]


_vp = C.c_void_p
_FAST = {}


def _fast(lib, name, *argtypes):
    """The same exported function with untyped pointer arguments (addresses as ints)."""
    f = _FAST.get(name)
    if f is None:
        f = _FAST[name] = C.CFUNCTYPE(C.c_int, *argtypes)((name, lib))
    return f


def spline_arrays(path) -> List[np.ndarray]:
    """Knots and CubicSpline1D coefficients of a CubicSpline2D-like object
    (reference attributes: .s, .sx.{a,b,c,d}, .sy.{a,b,c,d}; cubic_spline.py:30-45, 201-204)."""
    s = np.ascontiguousarray(np.asarray(path.s, dtype=np.float64))
    out = [s]
    for sp1 in (path.sx, path.sy):
        for name in "abcd":
            out.append(np.ascontiguousarray(np.asarray(getattr(sp1, name), dtype=np.float64)))
    n = len(s)
    want = [n, n, n - 1, n, n - 1, n, n - 1, n, n - 1]
    for arr, w in zip(out, want):
        if arr.shape != (w,):
            raise ValueError(f"spline coefficient array has shape {arr.shape}, expected ({w},)")
    return out


class BatchResult:
    """Results of one batch: the raw ``fot_result`` records plus convenient views."""

    def __init__(self, records, n: int, with_stop: Sequence[bool]):
        self.records = records
        self.n = n
        self._with_stop = list(with_stop)
        self._paths = None

    def __len__(self):
        return self.n

    def status(self, i: int) -> int:
        return int(self.records[i].status)

    def stats(self, i: int) -> Optional[Dict[str, int]]:
        """``last_check_stats`` of instance i (None where the reference leaves it None)."""
        r = self.records[i]
        if not r.stats_valid:
            return None
        d = {_abi.STATUS_NAMES[k]: int(r.stats[k]) for k in range(7)}
        if self._with_stop[i]:
            d["stop_distance_error"] = int(r.stats[7])
        return d

    def _path_block(self) -> np.ndarray:
        """[n, 15, FOT_MAX_NT] float64 view of the path arrays of all records (no copy)."""
        if self._paths is None:
            raw = np.frombuffer(self.records, dtype=np.uint8).reshape(-1, _abi.RESULT_BYTES)
            off = getattr(_abi.Result, _abi.PATH_FIELDS[0]).offset
            self._paths = raw[:, off:].view(np.float64).reshape(raw.shape[0], len(_abi.PATH_FIELDS), -1)
        return self._paths

    def path(self, i: int, as_arrays: bool = False) -> Optional[FrenetPath]:
        """The selected path of instance i as the reference's FrenetPath (lists); as_arrays=True: NumPy rows, for
        callers that take thousands of paths per second (closed_loop.py)."""
        r = self.records[i]
        if r.status != _abi.PLAN_OK:
            return None
        n = int(r.n_keep)
        block = self._path_block()[i, :, :n]
        fp = FrenetPath()
        for j, f in enumerate(_abi.PATH_FIELDS):
            setattr(fp, f, block[j].copy() if as_arrays else block[j].tolist())
        fp.cost = float(r.cost)
        return fp


class BatchPlanner:
    """One libfot handle: planner parameters + reference path on one GPU."""

    def __init__(self, reference_path=None, waypoints=None, device: int = -1, **planner_kwargs):
        self._lib = _abi.lib()
        fp = planner_kwargs.pop("footprint", None)
        if fp is not None:
            planner_kwargs["footprint_offsets"] = np.asarray(fp.offsets, dtype=float).tolist()
            planner_kwargs["footprint_radius"] = float(fp.radius)
        self.params = make_params(**planner_kwargs)
        h = C.c_void_p()
        _abi.check(None, self._lib.fot_create(C.byref(self.params), int(device), C.byref(h)))
        self._h = h
        _abi.register_owner(self)                # closed by _abi's atexit hook if the caller never does
        if reference_path is not None:
            self.set_path(reference_path)
        elif waypoints is not None:
            self.set_waypoints(*waypoints)

    # -- lifetime ---------------------------------------------------------
    def close(self):
        """Release the handle (idempotent).  ``fot_destroy`` polls the handle's streams for a bounded time and never
        blocks on a caller's stream; planners nobody closed are closed by ``_abi``'s atexit hook, which runs before
        the interpreter starts tearing modules down."""
        h, self._h = getattr(self, "_h", None), None
        if h:
            _abi.unregister_owner(self)
            self._lib.fot_destroy(h)

    def __del__(self):
        # Collected while the program runs: close.  During interpreter finalisation the atexit hook has already closed
        # every planner that was still registered (``_h`` is None then), so nothing calls into HIP from here.
        if sys.is_finalizing():
            return
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    # -- reference path ---------------------------------------------------
    def set_path(self, path):
        arrs = spline_arrays(path)
        _abi.check(self._h, self._lib.fot_set_path_coeffs(self._h, len(arrs[0]), *[_as_dp(a) for a in arrs]))

    def set_waypoints(self, wx, wy):
        wx = np.ascontiguousarray(wx, dtype=np.float64)
        wy = np.ascontiguousarray(wy, dtype=np.float64)
        if wx.shape != wy.shape or wx.ndim != 1:
            raise ValueError("waypoints must be two 1-D arrays of equal length")
        _abi.check(self._h, self._lib.fot_set_path_waypoints(self._h, len(wx), _as_dp(wx), _as_dp(wy)))

    def path_coeffs(self) -> List[np.ndarray]:
        n = C.c_int32(0)
        _abi.check(self._h, self._lib.fot_get_path_coeffs(self._h, C.byref(n), *([None] * 9)))
        k = n.value
        out = [np.zeros(k), np.zeros(k), np.zeros(k - 1), np.zeros(k), np.zeros(k - 1),
               np.zeros(k), np.zeros(k - 1), np.zeros(k), np.zeros(k - 1)]
        _abi.check(self._h, self._lib.fot_get_path_coeffs(self._h, C.byref(n), *[_as_dp(a) for a in out]))
        return out

    def spline_eval(self, s):
        s = np.ascontiguousarray(np.atleast_1d(s), dtype=np.float64)
        out = [np.zeros_like(s) for _ in range(5)]
        _abi.check(self._h, self._lib.fot_spline_eval(self._h, len(s), _as_dp(s), *[_as_dp(a) for a in out]))
        return out      # x, y, yaw, curvature, curvature_rate

    # -- planning ---------------------------------------------------------
    def plan_packed(self, pb: PackedBatch) -> BatchResult:
        out = (_abi.Result * max(pb.n, 1))()
        _abi.check(self._h, self._lib.fot_plan_batch(self._h, C.byref(pb.c), out))
        return BatchResult(out, pb.n, [not np.isnan(v) for v in pb.max_stop[: pb.n]])

    def plan_batch(self, requests: Sequence[PlanRequest], obstacle_dtype=np.float64) -> BatchResult:
        return self.plan_packed(PackedBatch(requests, obstacle_dtype))

    def plan_packed_device(self, batch_struct: _abi.Batch, out_dev_ptr: int, stream: Optional[int] = None):
        """Obstacles and results resident in HBM; enqueues on ``stream`` (a hipStream_t handle) and returns immediately.
        ``None`` / 0 = the handle's own stream, which no other stream waits for (torch's default stream has handle 0:
        pass an explicit ``torch.cuda.Stream`` and follow up on that stream, or call ``synchronize()``)."""
        _abi.check(self._h, self._lib.fot_plan_batch_device(self._h, C.byref(batch_struct), C.c_void_p(out_dev_ptr),
                                                            C.c_void_p(stream) if stream else None))

    @property
    def n_total_samples(self) -> int:
        """round(max_t / dt) + 1: samples per full-length candidate (the wire form's array length)."""
        return int(self._lib.fot_wire_n_total(self._h))

    def pack_records_device(self, n: int, records_dev_ptr: int, wire_dev_ptr: int, stream: Optional[int] = None):
        """fot_result[n] in HBM -> compact wire records in HBM (fot_pack_records_device), enqueued on ``stream``."""
        _abi.check(self._h, self._lib.fot_pack_records_device(self._h, int(n), C.c_void_p(records_dev_ptr),
                                                              C.c_void_p(wire_dev_ptr),
                                                              C.c_void_p(stream) if stream else None))

    # -- array interfaces: whole batches as NumPy arrays, no per-request Python objects (closed_loop.py) -----------
    RESULT_DT = np.dtype(_abi.Result)
    EGO_DT = np.dtype(_abi.Ego)
    SAFETY_DT = np.dtype(_abi.Safety)

    def plan_arrays(self, ego: np.ndarray, target_speed: np.ndarray, overrides: np.ndarray, max_stop: np.ndarray,
                    static_xy: Optional[np.ndarray], static_off: Optional[np.ndarray], dyn_xy: Optional[np.ndarray],
                    dyn_off: Optional[np.ndarray], dyn_dims: Optional[np.ndarray]) -> np.ndarray:
        """n plan() calls given column-wise (``fot_plan_batch`` with host arrays): ``ego`` structured [n] of ``EGO_DT``
        (has_prev_s 2 = chained on the request before), ``overrides`` [n, 4] float64 (NaN = key absent; max_speed,
        max_accel, max_curvature, max_lat_accel), ``max_stop`` [n] (NaN = None), obstacles as in ``fot_batch`` (float64).
        Returns the records as a structured array of ``RESULT_DT``."""
        n = int(ego.shape[0])
        out = np.zeros(max(n, 1), dtype=self.RESULT_DT)
        if n == 0:
            return out[:0]
        b = _abi.Batch()
        b.n_inst = n
        b.obstacle_dtype = _abi.F64
        ego = np.ascontiguousarray(ego, dtype=self.EGO_DT)
        tgt = np.ascontiguousarray(target_speed, dtype=np.float64)
        ov = np.ascontiguousarray(overrides, dtype=np.float64).reshape(n, 4)
        ms = np.ascontiguousarray(max_stop, dtype=np.float64)
        b.ego = C.cast(_addr(ego), C.POINTER(_abi.Ego))
        b.target_speed = C.cast(_addr(tgt), _dp)
        b.overrides = C.cast(_addr(ov), C.POINTER(_abi.Overrides))
        b.max_stop_distance = C.cast(_addr(ms), _dp)
        keep = [ego, tgt, ov, ms]
        if static_xy is not None and static_off is not None and len(static_xy):
            sx = np.ascontiguousarray(static_xy, dtype=np.float64)
            so = np.ascontiguousarray(static_off, dtype=np.int32)
            b.static_xy, b.static_off = _addr(sx), C.cast(_addr(so), _ip)
            keep += [sx, so]
        if dyn_xy is not None and dyn_off is not None and len(dyn_xy):
            dx = np.ascontiguousarray(dyn_xy, dtype=np.float64)
            do = np.ascontiguousarray(dyn_off, dtype=np.int64)
            dd = np.ascontiguousarray(dyn_dims, dtype=np.int32).reshape(n, 4)
            b.dyn_xy, b.dyn_off = _addr(dx), C.cast(_addr(do), C.POINTER(C.c_int64))
            b.dyn_dims = C.cast(_addr(dd), _ip)
            keep += [dx, do, dd]
        f = _fast(self._lib, "fot_plan_batch", _vp, _vp, _vp)
        _abi.check(self._h, f(self._h, C.addressof(b), _addr(out)))
        return out

    def safety_metrics_cat(self, egos: np.ndarray, ped_off: np.ndarray, ped_pos: np.ndarray, ped_vel: np.ndarray,
                           ego_radius: float, ped_radius: float, use_footprint: bool = True) -> np.ndarray:
        """``safety_metrics`` with the pedestrians of all egos concatenated: ego i owns rows [ped_off[i], ped_off[i+1])."""
        ego = np.ascontiguousarray(egos, dtype=np.float64).reshape(-1, 4)
        n = ego.shape[0]
        off = np.ascontiguousarray(ped_off, dtype=np.int32)
        pos = np.ascontiguousarray(ped_pos, dtype=np.float64)
        vel = np.ascontiguousarray(ped_vel, dtype=np.float64)
        out = np.zeros(max(n, 1), dtype=self.SAFETY_DT)
        f = _fast(self._lib, "fot_safety_metrics_batch", _vp, C.c_int32, _vp, _vp, _vp, _vp, C.c_double, C.c_double,
                  C.c_int32, _vp)
        _abi.check(self._h, f(self._h, n, _addr(ego), _addr(off), _addr(pos) if pos.size else None,
                              _addr(vel) if vel.size else None, float(ego_radius), float(ped_radius),
                              int(bool(use_footprint)), _addr(out)))
        return out[:n]

    def nearest_s_arrays(self, x, y, yaw, v, a, prev_s) -> np.ndarray:
        """new_prev_s of ``fot_frenet_state_batch`` for n egos given column-wise (prev_s NaN = no cached arc length)."""
        n = len(x)
        ego = np.zeros(max(n, 1), dtype=self.EGO_DT)
        ego["x"][:n], ego["y"][:n], ego["yaw"][:n], ego["v"][:n], ego["a"][:n] = x, y, yaw, v, a
        ps = np.asarray(prev_s, dtype=np.float64)
        ego["has_prev_s"][:n] = ~np.isnan(ps)
        ego["prev_s"][:n] = np.where(np.isnan(ps), 0.0, ps)
        nps = np.zeros(max(n, 1))
        f = _fast(self._lib, "fot_frenet_state_batch", _vp, C.c_int32, _vp, _vp, _vp, _vp, _vp)
        _abi.check(self._h, f(self._h, n, _addr(ego), None, None, _addr(nps), None))
        return nps[:n]

    # -- one closed-loop step in two calls (fot_loop_*): the prediction tensor stays in HBM -------------------------
    LOOP_REQUEST_DT = np.dtype(_abi.LoopRequest)

    def loop_set_static(self, points: Optional[np.ndarray]) -> None:
        """The static obstacle points every request of the loop sees (``fot_loop_set_static``)."""
        pts = np.ascontiguousarray(np.empty((0, 2)) if points is None else points, dtype=np.float64).reshape(-1, 2)
        _abi.check(self._h, self._lib.fot_loop_set_static(self._h, len(pts), _addr(pts) if len(pts) else None))

    def loop_plan(self, requests: np.ndarray, frame: Optional[dict] = None, view: bool = False):
        """``fot_loop_plan``: constant-velocity prediction of the frame's pedestrians into the handle's own tensor,
        safety metrics of the current ego states, and the requests' plan() calls against that tensor -- one
        synchronisation.  ``requests``: structured [r] of ``LOOP_REQUEST_DT``.  ``frame`` (None = the tensor of the
        previous call): ped_off [n+1], ped_pos / ped_vel [sum P, 2], obs_last / obs_prev [sum P, 2] float32 or None
        (predictor not ready), prepend [n] bool, ego [n, 4] (x, y, yaw, v) or None, staleness, pred_len, rp
        (``_abi.ResampleParams``), ego_radius, ped_radius, use_footprint.
        Returns (records, metrics): the records as a structured array of ``RESULT_DT`` -- a COPY by default; with
        ``view=True`` a view of the handle's pinned block, valid only until the next ``loop_plan`` or ``close()`` on this
        planner (the lock-step driver reads it at once and keeps nothing) -- and the metrics [n] of ``SAFETY_DT`` (None
        without a frame or egos)."""
        req = np.ascontiguousarray(requests, dtype=self.LOOP_REQUEST_DT)
        r = int(req.shape[0])
        fr_addr, metrics, keep = None, None, []
        if frame is not None:
            f = _abi.LoopFrame()
            off = np.ascontiguousarray(frame["ped_off"], dtype=np.int32)
            n = len(off) - 1
            f.n_episodes, f.pred_len = n, int(frame.get("pred_len", 1))
            f.use_footprint = int(bool(frame.get("use_footprint", True)))
            pos = np.ascontiguousarray(frame["ped_pos"], dtype=np.float64)
            vel = np.ascontiguousarray(frame["ped_vel"], dtype=np.float64)
            f.ped_off, f.ped_pos, f.ped_vel = _addr(off), _addr(pos) if pos.size else None, _addr(vel) if vel.size else None
            keep += [off, pos, vel]
            if frame.get("obs_last") is not None:
                last = np.ascontiguousarray(frame["obs_last"], dtype=np.float32)
                pre = np.ascontiguousarray(frame["prepend"], dtype=np.uint8)
                f.obs_last, f.prepend = _addr(last), _addr(pre) if pre.size else None
                keep += [last, pre]
                if frame.get("obs_prev") is not None:
                    prev = np.ascontiguousarray(frame["obs_prev"], dtype=np.float32)
                    f.obs_prev = _addr(prev)
                    keep.append(prev)
                f.rp = frame["rp"]
            if frame.get("ego") is not None:
                ego = np.ascontiguousarray(frame["ego"], dtype=np.float64).reshape(n, 4)
                f.ego = _addr(ego)
                keep.append(ego)
                metrics = np.zeros(max(n, 1), dtype=self.SAFETY_DT)
            f.staleness = float(frame.get("staleness", 0.0))
            f.ego_radius, f.ped_radius = float(frame["ego_radius"]), float(frame["ped_radius"])
            fr_addr = C.addressof(f)
        rec_ptr = C.c_void_p(0)
        fn = _fast(self._lib, "fot_loop_plan", _vp, _vp, C.c_int32, _vp, _vp, _vp)
        _abi.check(self._h, fn(self._h, fr_addr, r, _addr(req) if r else None,
                               _addr(metrics) if metrics is not None else None, C.addressof(rec_ptr)))
        if r and rec_ptr.value:
            buf = (C.c_char * (r * _abi.RESULT_BYTES)).from_address(rec_ptr.value)
            records = np.frombuffer(buf, dtype=self.RESULT_DT, count=r)
            if not view:
                records = records.copy()
        else:
            records = np.zeros(0, dtype=self.RESULT_DT)
        return records, (metrics[:n] if metrics is not None else None)

    def loop_begin(self, config: "_abi.LoopConfig", ego5: np.ndarray) -> None:
        """``fot_loop_begin``: n episode slots (ego5 [n, 5] = x, y, yaw, v, a), every fail-safe machine NORMAL."""
        ego = np.ascontiguousarray(ego5, dtype=np.float64).reshape(-1, 5)
        self._loop_n = len(ego)
        _abi.check(self._h, self._lib.fot_loop_begin(self._h, len(ego), C.addressof(config), _addr(ego) if len(ego) else None))

    def loop_step(self, frame: dict, episode: np.ndarray) -> dict:
        """``fot_loop_step``: the whole lock step of the frame's episodes (``episode[i]`` = slot of episode i) in one
        call -- prediction, metrics, level-0 plans, escalation levels of the failed ones, the retry loop, ego update /
        emergency stop, metrics of the new states and their nearest path point.  ``frame`` as for ``loop_plan`` (no
        ``ego``).  Returns the per-episode outputs as arrays plus ``records``: a structured VIEW of the step's records in
        the handle's pinned block, valid until the next loop call."""
        f, keep = self._loop_frame(frame)
        ep = np.ascontiguousarray(episode, dtype=np.int32)
        n = len(ep)
        o = dict(ego=np.zeros((n, 5)), jerk=np.zeros(n), state=np.zeros(n, np.int32), stats=np.zeros((n, 8), np.int32),
                 record=np.zeros(n, np.int32), keep=np.zeros(n, np.int32), cost=np.zeros(n),
                 before=np.zeros(max(n, 1), dtype=self.SAFETY_DT), after=np.zeros(max(n, 1), dtype=self.SAFETY_DT),
                 s_now=np.zeros(n))
        so = _abi.LoopStepOut()
        for name, arr in o.items():
            setattr(so, name, _addr(arr) if arr.size else None)
        fn = _fast(self._lib, "fot_loop_step", _vp, _vp, _vp, _vp)
        _abi.check(self._h, fn(self._h, C.addressof(f), _addr(ep) if n else None, C.addressof(so)))
        if so.n_records and so.records:
            buf = (C.c_char * (so.n_records * _abi.RESULT_BYTES)).from_address(so.records)
            o["records"] = np.frombuffer(buf, dtype=self.RESULT_DT, count=so.n_records)
        else:
            o["records"] = np.zeros(0, dtype=self.RESULT_DT)
        o["before"], o["after"] = o["before"][:n], o["after"][:n]
        return o

    def _loop_frame(self, frame: dict):
        """fot_loop_frame from the dictionary ``loop_plan`` / ``loop_step`` take (+ the arrays it points into)."""
        f, keep = _abi.LoopFrame(), []
        off = np.ascontiguousarray(frame["ped_off"], dtype=np.int32)
        n = len(off) - 1
        f.n_episodes, f.pred_len = n, int(frame.get("pred_len", 1))
        f.use_footprint = int(bool(frame.get("use_footprint", True)))
        pos = np.ascontiguousarray(frame["ped_pos"], dtype=np.float64)
        vel = np.ascontiguousarray(frame["ped_vel"], dtype=np.float64)
        f.ped_off, f.ped_pos, f.ped_vel = _addr(off), _addr(pos) if pos.size else None, _addr(vel) if vel.size else None
        keep += [off, pos, vel]
        if frame.get("obs_last") is not None:
            last = np.ascontiguousarray(frame["obs_last"], dtype=np.float32)
            pre = np.ascontiguousarray(frame["prepend"], dtype=np.uint8)
            f.obs_last, f.prepend = _addr(last), _addr(pre) if pre.size else None
            keep += [last, pre]
            if frame.get("obs_prev") is not None:
                prev = np.ascontiguousarray(frame["obs_prev"], dtype=np.float32)
                f.obs_prev = _addr(prev)
                keep.append(prev)
            f.rp = frame["rp"]
        if frame.get("dist_raw") is not None:                          # raw samples of a multi-sample predictor, in HBM
            f.dist_raw, f.dist_S, f.dist_dtype = int(frame["dist_raw"]), int(frame["dist_S"]), int(frame["dist_dtype"])
        f.staleness = float(frame.get("staleness", 0.0))
        f.ego_radius, f.ped_radius = float(frame["ego_radius"]), float(frame["ped_radius"])
        f._keep = keep                                               # (the arrays live as long as the structure)
        return f, keep

    def gather_paths(self, records: np.ndarray, index: np.ndarray, kmax: int, out: Optional[np.ndarray] = None) -> np.ndarray:
        """The first ``kmax`` samples of the 15 path arrays of ``records[index]`` as one dense [15, n, kmax] block
        (``fot_gather_paths``; ``_abi.PATH_FIELDS`` order), written into ``out`` (C-contiguous, that shape) if given."""
        idx = np.ascontiguousarray(index, dtype=np.int32)
        shape = (len(_abi.PATH_FIELDS), len(idx), int(kmax))
        if out is None:
            out = np.empty(shape)
        elif out.shape != shape or out.dtype != np.float64 or not out.flags.c_contiguous:
            raise ValueError(f"out must be a C-contiguous float64 array of shape {shape}")
        if out.size:
            fn = _fast(self._lib, "fot_gather_paths", _vp, C.c_int32, _vp, C.c_int32, _vp)
            rc = fn(_addr(records), len(idx), _addr(idx), int(kmax), _addr(out))
            if rc != 0:
                raise _abi.FotError(rc, "fot_gather_paths: bad index or length")
        return out

    def loop_observe(self, ego5: np.ndarray, prev_s: np.ndarray):
        """``fot_loop_observe``: safety metrics of the new ego states [n, 5] (x, y, yaw, v, a; ego i = episode i of
        the frame) against the frame's pedestrians, and the arc length of their nearest path point (prev_s NaN = no
        cached arc length) -- one synchronisation.  Returns (metrics [n] of ``SAFETY_DT``, s [n])."""
        ego = np.ascontiguousarray(ego5, dtype=np.float64).reshape(-1, 5)
        n = ego.shape[0]
        ps = np.ascontiguousarray(prev_s, dtype=np.float64)
        metrics = np.zeros(max(n, 1), dtype=self.SAFETY_DT)
        s = np.zeros(max(n, 1))
        fn = _fast(self._lib, "fot_loop_observe", _vp, C.c_int32, _vp, _vp, _vp, _vp)
        _abi.check(self._h, fn(self._h, n, _addr(ego), _addr(ps), _addr(metrics), _addr(s)))
        return metrics[:n], s[:n]

    def loop_observe_begin(self, ego5: np.ndarray, prev_s: np.ndarray):
        """``loop_observe`` in two halves: enqueues the launches and returns a callable that waits for them and returns
        (metrics, s) -- host work in between overlaps with the device.  No other call on this planner before it."""
        ego = np.ascontiguousarray(ego5, dtype=np.float64).reshape(-1, 5)
        n = ego.shape[0]
        ps = np.ascontiguousarray(prev_s, dtype=np.float64)
        fn = _fast(self._lib, "fot_loop_observe_begin", _vp, C.c_int32, _vp, _vp)
        _abi.check(self._h, fn(self._h, n, _addr(ego), _addr(ps)))

        def collect():
            metrics = np.zeros(max(n, 1), dtype=self.SAFETY_DT)
            s = np.zeros(max(n, 1))
            fe = _fast(self._lib, "fot_loop_observe_end", _vp, _vp, _vp)
            _abi.check(self._h, fe(self._h, _addr(metrics), _addr(s)))
            return metrics[:n], s[:n]
        return collect

    def synchronize(self):
        _abi.check(self._h, self._lib.fot_synchronize(self._h))

    def profile(self, on: bool):
        """Bracket every kernel launch with HIP events on its stream (fot_profile_enable)."""
        _abi.check(self._h, self._lib.fot_profile_enable(self._h, 1 if on else 0))

    def profile_read(self, reset: bool = True) -> Dict[str, Dict[str, float]]:
        """Per kernel: launches and summed device ms since the last reset (waits for the work)."""
        n = _abi.PROFILE_KERNELS
        launches = np.zeros(n, np.int32)
        ms = np.zeros(n)
        rc = self._lib.fot_profile_read(self._h, 1 if reset else 0, n, launches.ctypes.data_as(_ip), _as_dp(ms))
        if rc < 0:
            _abi.check(self._h, rc)
        return {self._lib.fot_profile_kernel_name(k).decode(): {"launches": int(launches[k]), "total_ms": float(ms[k])}
                for k in range(n)}

    def candidates(self, inst: int = 0, cap: int = 1 << 16):
        """Per-candidate (cost, status, keep, n_t) of an instance of the last plan call."""
        cost = np.zeros(cap)
        status = np.zeros(cap, np.int32)
        keep = np.zeros(cap, np.int32)
        nt = np.zeros(cap, np.int32)
        n = self._lib.fot_debug_candidates(self._h, inst, cap, _as_dp(cost), status.ctypes.data_as(_ip),
                                           keep.ctypes.data_as(_ip), nt.ctypes.data_as(_ip))
        if n < 0:
            _abi.check(self._h, n)
        n = min(n, cap)
        return cost[:n], status[:n], keep[:n], nt[:n]

    def margins(self, inst: int = 0, cap: int = 1 << 16) -> np.ndarray:
        """[n_cand, 8] smallest relative distance to a threshold per candidate and decision group
        (``_abi.MARGIN_NAMES``) of an instance of the last plan call; +inf where no such decision was made."""
        out = np.full((cap, _abi.MARGIN_GROUPS), np.inf)
        n = self._lib.fot_debug_margins(self._h, inst, cap, _as_dp(out))
        if n < 0:
            _abi.check(self._h, n)
        return out[: min(n, cap)]

    def set_eval_segments(self, n_seg: int) -> None:
        """Test hook (``fot_debug_set_eval_segments``): time segments per candidate in the evaluation kernel,
        1..4 forced, 0 = chosen by batch size."""
        _abi.check(self._h, self._lib.fot_debug_set_eval_segments(self._h, int(n_seg)))

    def set_tile_cut(self, cut) -> None:
        """Test hook (``fot_debug_set_tile_cut``): 0 / "auto", 1 / "wave" (k_evaluate, per-wave rows), 2 / "group"
        (k_evaluate_group, four tiles per row table)."""
        cut = {"auto": 0, "wave": 1, "group": 2}.get(cut, cut)
        _abi.check(self._h, self._lib.fot_debug_set_tile_cut(self._h, int(cut)))

    def time_info(self, time: float):
        """(n_t, quartic inverse [2, 2], quintic inverse [3, 3]) the library solves a horizon of ``time`` seconds with
        (``fot_debug_time_info``)."""
        n_t = C.c_int32(0)
        qa, qi = np.zeros(4), np.zeros(9)
        _abi.check(self._h, self._lib.fot_debug_time_info(self._h, float(time), C.byref(n_t), _as_dp(qa), _as_dp(qi)))
        return int(n_t.value), qa.reshape(2, 2), qi.reshape(3, 3)

    def candidate_path(self, index: int, inst: int = 0) -> FrenetPath:
        """Candidate ``index`` of the last plan call as generated + converted, before truncation."""
        arr = np.zeros((15, _abi.MAX_NT))
        nt = C.c_int32(0)
        _abi.check(self._h, self._lib.fot_debug_candidate_path(self._h, inst, int(index), _as_dp(arr), C.byref(nt)))
        fp = FrenetPath()
        for k, f in enumerate(_abi.PATH_FIELDS):
            setattr(fp, f, arr[k, : nt.value].tolist())
        return fp

    def frenet_states(self, egos: Sequence[PlanRequest]):
        n = len(egos)
        arr = (_abi.Ego * max(n, 1))()
        for i, r in enumerate(egos):
            e = arr[i]
            e.x, e.y, e.yaw, e.v, e.a, e.last_kappa = r.x, r.y, r.yaw, r.v, r.a, r.last_kappa
            e.has_prev_s = 0 if r.prev_s is None else 1
            e.prev_s = 0.0 if r.prev_s is None else r.prev_s
        fr = np.zeros((max(n, 1), 6))
        ref = np.zeros((max(n, 1), 6))
        nps = np.zeros(max(n, 1))
        ok = np.zeros(max(n, 1), np.int32)
        _abi.check(self._h, self._lib.fot_frenet_state_batch(self._h, n, arr, _as_dp(fr), _as_dp(ref), _as_dp(nps),
                                                             ok.ctypes.data_as(_ip)))
        return fr[:n], ref[:n], nps[:n], ok[:n]

    def safety_metrics(self, egos, ped_positions: Sequence, ped_velocities: Sequence, ego_radius: float,
                       ped_radius: float, use_footprint: bool = True) -> np.ndarray:
        """compute_safety_metrics_static (data_structures.py:301-388) for n egos in one launch.

        egos [n, 4] = x, y, yaw, v; ped_positions / ped_velocities: one [P_i, 2] array per ego.  Returns a structured
        array with fields min_distance, ttc, clearance, clearance_ahead, collision."""
        ego = np.ascontiguousarray(egos, dtype=np.float64).reshape(-1, 4)
        n = ego.shape[0]
        if len(ped_positions) != n or len(ped_velocities) != n:
            raise ValueError("one pedestrian array per ego is required")
        off = np.zeros(n + 1, np.int32)
        pos, vel = [], []
        for i in range(n):
            p = np.asarray(ped_positions[i], dtype=np.float64).reshape(-1, 2)
            v = np.asarray(ped_velocities[i], dtype=np.float64).reshape(-1, 2)
            if p.shape != v.shape:
                raise ValueError(f"ego {i}: positions {p.shape} and velocities {v.shape} differ")
            off[i + 1] = off[i] + p.shape[0]
            pos.append(p); vel.append(v)
        pos = np.ascontiguousarray(np.concatenate(pos, axis=0)) if n else np.empty((0, 2))
        vel = np.ascontiguousarray(np.concatenate(vel, axis=0)) if n else np.empty((0, 2))
        out = (_abi.Safety * max(n, 1))()
        _abi.check(self._h, self._lib.fot_safety_metrics_batch(
            self._h, n, _as_dp(ego), off.ctypes.data_as(_ip), _as_dp(pos) if pos.size else None,
            _as_dp(vel) if vel.size else None, float(ego_radius), float(ped_radius), int(bool(use_footprint)), out))
        return np.ctypeslib.as_array(out)[:n].copy() if n else np.zeros(0, dtype=np.dtype(_abi.Safety))

    @staticmethod
    def _pack_paths(paths: Sequence, fields: Sequence[str]):
        """FrenetPath-like objects -> {field: [n, MAX_NT]} arrays, lengths, presence flags."""
        n = len(paths)
        arrs = {f: np.zeros((max(n, 1), _abi.MAX_NT)) for f in fields}
        ln = np.zeros(max(n, 1), np.int32)
        flags = np.zeros(max(n, 1), np.int32)
        for i, fp in enumerate(paths):
            m = min(len(fp.x), len(fp.t))                        # frenet_planner.py:1146
            if m > _abi.MAX_NT:
                raise ValueError(f"path longer than {_abi.MAX_NT} samples")
            ln[i] = m
            present = {}
            for f in fields:
                v = getattr(fp, f, None)
                v = np.zeros(0) if v is None else np.asarray(v, dtype=float)
                present[f] = len(v) >= m and m > 0
                if f == "yaw" and 0 < len(v) < m:                # :1158-1161 hold the last value
                    v = np.concatenate([v, np.full(m - len(v), v[-1])])
                k = min(len(v), m)
                arrs[f][i, :k] = v[:k]
            geo = all(present.get(f, False) for f in ("x", "y", "yaw", "s", "d") if f in fields)
            flags[i] = (1 if geo else 0) | (2 if present.get("d", False) else 0)
        return arrs, ln, flags

    def _obstacle_args(self, static, dyn, dist):
        req = PlanRequest(0, 0, 0, 0, 0, static=static, dyn=dyn, dist=dist)
        pb = PackedBatch([req], np.float64)
        mode, S, P, T = (int(v) for v in pb.dyn_dims[0])
        st = pb.static_xy
        return pb, (int(st.shape[0]), _as_dp(st) if st.size else None, mode, S, P, T,
                    _as_dp(pb.dyn_xy) if pb.dyn_xy.size else None)

    def paths_collision_free(self, paths: Sequence, static=None, dyn=None, dist=None) -> np.ndarray:
        """_path_is_collision_free (frenet_planner.py:1035-1047) for FrenetPath-like objects."""
        n = len(paths)
        arrs, ln, _ = self._pack_paths(paths, ("x", "y", "yaw", "t"))
        keep, oargs = self._obstacle_args(static, dyn, dist)
        free = np.zeros(max(n, 1), np.int32)
        _abi.check(self._h, self._lib.fot_check_collision_paths(
            self._h, n, ln.ctypes.data_as(_ip), _as_dp(arrs["x"]), _as_dp(arrs["y"]), _as_dp(arrs["yaw"]),
            _as_dp(arrs["t"]), *oargs, free.ctypes.data_as(_ip)))
        return free[:n].astype(bool)

    def check_paths(self, paths: Sequence, static=None, dyn=None, overrides=None, dist=None,
                    max_stop_distance=None) -> np.ndarray:
        """_check_paths (+ stop-distance filter) categories of FrenetPath-like objects: FOT_ST_* per path."""
        n = len(paths)
        fields = ("x", "y", "yaw", "v", "a", "c", "d", "s", "t")
        arrs, ln, flags = self._pack_paths(paths, fields)
        for i, fp in enumerate(paths):                           # :933-940 silently skipped
            if len(fp.x) == 0 or len(fp.x) != len(fp.t):
                ln[i] = 0
        keep, oargs = self._obstacle_args(static, dyn, dist)
        ov = _abi.Overrides()
        o = overrides or {}
        nan = float("nan")
        ov.max_speed, ov.max_accel = float(o.get("max_speed", nan)), float(o.get("max_accel", nan))
        ov.max_curvature, ov.max_lat_accel = float(o.get("max_curvature", nan)), float(o.get("max_lat_accel", nan))
        status = np.zeros(max(n, 1), np.int32)
        _abi.check(self._h, self._lib.fot_check_paths(
            self._h, n, ln.ctypes.data_as(_ip), flags.ctypes.data_as(_ip), *[_as_dp(arrs[f]) for f in fields],
            C.byref(ov), nan if max_stop_distance is None else float(max_stop_distance), *oargs,
            status.ctypes.data_as(_ip)))
        status = status[:n]
        status[ln[:n] == 0] = _abi.ST_DROPPED
        return status


class _NearestPointState:
    """Stands in for ``planner.converter`` of the reference: carries the cached arc length
    (CoordinateConverter._prev_s, coordinate_converter.py:221, 283); absent until the first search."""

    def __init__(self, path):
        self.reference_path = path


class FrenetPlanner:
    """Drop-in for the reference's FrenetPlanner (frenet_planner.py:125-304) backed by libfot."""

    def __init__(self, reference_path, max_speed: float = MAX_SPEED, max_accel: float = MAX_ACCEL,
                 max_curvature: float = MAX_CURVATURE, dt: float = DT, d_road_w: float = D_ROAD_W,
                 max_road_width: float = MAX_ROAD_WIDTH, robot_radius: float = ROBOT_RADIUS,
                 obstacle_radius: float = 0.3, min_t: float = MIN_T, max_t: float = MAX_T, d_t_s: float = D_T_S,
                 n_s_sample: int = N_S_SAMPLE, **kwargs):
        self.csp = reference_path
        self.max_speed = max_speed
        self.max_accel = max_accel
        self.max_curvature = max_curvature
        self.max_lat_accel = float(kwargs.get("max_lat_accel", 3.0))
        self.dt = dt
        self.d_road_w = d_road_w
        self.max_road_width = max_road_width
        self.robot_radius = robot_radius
        self.obstacle_radius = obstacle_radius
        self.min_t = min_t
        self.max_t = max_t
        self.d_t_s = d_t_s
        self.n_s_sample = n_s_sample
        self.k_j = kwargs.get("k_j", 0.1)
        self.k_t = kwargs.get("k_t", 0.1)
        self.k_d = kwargs.get("k_d", 1.0)
        self.k_s_dot = kwargs.get("k_s_dot", 1.0)
        self.k_lat = kwargs.get("k_lat", 1.0)
        self.k_lon = kwargs.get("k_lon", 1.0)
        self.chance_epsilon = float(kwargs.get("chance_epsilon", 0.0))
        self.collision_margin_inflation = float(kwargs.get("collision_margin_inflation", 1.0))
        self.footprint = kwargs.get("footprint", None)
        self.converter = _NearestPointState(reference_path)
        self._last_kappa = 0.0
        self.last_check_stats = None
        # The reference plans any lattice and its plan() never raises (frenet_planner.py:266-268: a failure is `None`).
        # The library has capacities (include/fot.h FOT_MAX_*: 256 samples per candidate, 64 horizons, 32 terminal speeds,
        # 32 brake horizons, 8 footprint circles, 64 prediction samples); a configuration or a call beyond them keeps the
        # reference's contract: a warning once, plan() returns None with last_check_stats None, and `last_error` says why.
        self.last_error: Optional[str] = None
        self._warned = False
        try:
            self._engine = BatchPlanner(
                reference_path=reference_path, device=int(kwargs.get("device", -1)),
                max_speed=max_speed, max_accel=max_accel, max_curvature=max_curvature, dt=dt, d_road_w=d_road_w,
                max_road_width=max_road_width, robot_radius=robot_radius, obstacle_radius=obstacle_radius, min_t=min_t,
                max_t=max_t, d_t_s=d_t_s, max_lat_accel=self.max_lat_accel, k_j=self.k_j, k_t=self.k_t, k_d=self.k_d,
                k_s_dot=self.k_s_dot, k_lat=self.k_lat, k_lon=self.k_lon, chance_epsilon=self.chance_epsilon,
                collision_margin_inflation=self.collision_margin_inflation, footprint=self.footprint)
        except _abi.FotError as e:
            if e.code != _abi.ERR_UNSUPPORTED:
                raise
            self._engine = None
            self._unsupported(str(e))

    def _unsupported(self, msg: str) -> None:
        self.last_error = msg
        if not self._warned:
            import warnings
            warnings.warn(f"FrenetPlanner: beyond the library's capacities, plan() returns None ({msg})", RuntimeWarning,
                          stacklevel=3)
            self._warned = True

    @property
    def engine(self) -> BatchPlanner:
        return self._engine

    def _request(self, ego_state, static_obstacles, dynamic_obstacles, target_speed, constraint_overrides,
                 dynamic_obstacles_distribution, max_stop_distance) -> PlanRequest:
        return PlanRequest(
            x=float(ego_state.x), y=float(ego_state.y), yaw=float(ego_state.yaw), v=float(ego_state.v),
            a=float(ego_state.a), target_speed=float(target_speed), last_kappa=float(self._last_kappa),
            prev_s=getattr(self.converter, "_prev_s", None), overrides=constraint_overrides or None,
            max_stop_distance=max_stop_distance, static=static_obstacles, dyn=dynamic_obstacles,
            dist=dynamic_obstacles_distribution)

    def plan(self, ego_state: EgoVehicleState, static_obstacles: np.ndarray,
             dynamic_obstacles: Optional[np.ndarray] = None, target_speed: float = TARGET_SPEED,
             constraint_overrides: Optional[Dict[str, float]] = None,
             dynamic_obstacles_distribution: Optional[np.ndarray] = None,
             max_stop_distance: Optional[float] = None) -> Optional[FrenetPath]:
        """Same contract as the reference (frenet_planner.py:227-304): best path or None; never
        raises on "no path"; updates last_check_stats, _last_kappa and the nearest-point cache."""
        self.last_check_stats = None
        if self._engine is None:                                   # (a configuration beyond the library's capacities)
            return None
        req = self._request(ego_state, static_obstacles, dynamic_obstacles, target_speed, constraint_overrides,
                            dynamic_obstacles_distribution, max_stop_distance)
        try:
            res = self._engine.plan_batch([req])
        except _abi.FotError as e:                                 # (a call beyond them: more prediction samples than 64, ...)
            if e.code != _abi.ERR_UNSUPPORTED:
                raise
            self._unsupported(str(e))
            return None
        self.last_error = None
        rec = res.records[0]
        if not np.isnan(rec.new_prev_s):
            self.converter._prev_s = float(rec.new_prev_s)
        self.last_check_stats = res.stats(0)
        path = res.path(0)
        if path is not None:
            self._last_kappa = float(rec.new_last_kappa)
        return path

    def reset_ego_curvature(self):
        self._last_kappa = 0.0

    # -- stages the reference's tests reach into ---------------------------
    def _cartesian_to_frenet_state(self, ego_state) -> Optional[FrenetState]:
        req = PlanRequest(ego_state.x, ego_state.y, ego_state.yaw, ego_state.v, ego_state.a,
                          last_kappa=self._last_kappa, prev_s=getattr(self.converter, "_prev_s", None))
        fr, _ref, nps, ok = self._engine.frenet_states([req])
        if not np.isnan(nps[0]):
            self.converter._prev_s = float(nps[0])
        if not ok[0]:
            return None
        return FrenetState(*[float(v) for v in fr[0]])

    def _path_is_collision_free(self, fp, static_obstacles, dynamic_obstacles, dynamic_distribution) -> bool:
        return bool(self._engine.paths_collision_free([fp], static_obstacles, dynamic_obstacles,
                                                      dynamic_distribution)[0])

    def _check_collision(self, fp, static_obstacles, dynamic_obstacles=None) -> bool:
        return self._path_is_collision_free(fp, static_obstacles, dynamic_obstacles, None)

    def _engine_for_epsilon(self, epsilon: float) -> BatchPlanner:
        """chance_epsilon is a handle constant; other values get their own (cached) handle."""
        if float(epsilon) == self.chance_epsilon:
            return self._engine
        cache = self.__dict__.setdefault("_eps_engines", {})
        if float(epsilon) not in cache:
            p = self._engine.params
            kw = {name: getattr(p, name) for name, _ in p._fields_
                  if name not in ("_pad", "n_circles", "footprint_radius", "footprint_offsets", "chance_epsilon")}
            cache[float(epsilon)] = BatchPlanner(reference_path=self.csp, chance_epsilon=float(epsilon),
                                                 footprint=self.footprint, **kw)
        return cache[float(epsilon)]

    def _check_collision_distribution(self, fp, static_obstacles, dynamic_distribution, epsilon=None) -> bool:
        eng = self._engine_for_epsilon(self.chance_epsilon if epsilon is None else epsilon)
        if dynamic_distribution is None or np.size(dynamic_distribution) == 0:
            return bool(eng.paths_collision_free([fp], static_obstacles, None, None)[0])
        return bool(eng.paths_collision_free([fp], static_obstacles, None, dynamic_distribution)[0])

    def _check_paths(self, fp_list, static_obstacles, dynamic_obstacles=None, constraint_overrides=None,
                     dynamic_obstacles_distribution=None) -> dict:
        """Same categorisation as the reference's _check_paths (frenet_planner.py:891-993), on the device."""
        status = self._engine.check_paths(fp_list, static_obstacles, dynamic_obstacles, constraint_overrides,
                                          dynamic_obstacles_distribution)
        out = {k: [] for k in _abi.STATUS_NAMES[:7]}
        for fp, st in zip(fp_list, status):
            if st < 7:
                out[_abi.STATUS_NAMES[st]].append(fp)
        return out

    # The generation / conversion stages as separate calls (frenet_planner.py:376-503, 736-889).  libfot generates,
    # converts and checks a candidate in one pass over registers; these views re-run the lattice from the GIVEN Frenet
    # state (FOT_EGO_IS_FRENET) and read the candidates back through the debug entries -- what the reference's tests
    # look at, not a path plan() takes.
    _FRENET_FIELDS = ("t", "s", "s_d", "s_dd", "s_ddd", "d", "d_d", "d_dd", "d_ddd")
    _CART_FIELDS = ("x", "y", "yaw", "v", "a", "c")

    # --- the reference's polynomial builders, as views (its test-suite calls them; plan() never does: the lattice is
    # generated on the device).  The boundary-value inverses are the library's own closed forms.
    def _build_time_cache(self, time: float) -> "TimeCache":
        """frenet_planner.py:586-617: inclusive sample grid of the horizon, its powers, the two inverses."""
        n_t, qa, qi = self._engine.time_info(time)
        t = np.arange(n_t) * self.dt
        t2 = t * t
        return TimeCache(t=t, t2=t2, t3=t2 * t, t4=t2 * t2, t5=t2 * t2 * t, quartic_A_inv=qa, quintic_A_inv=qi)

    def _build_longitudinal_profiles(self, frenet_state, target_velocities, time: float, time_cache) -> list:
        """frenet_planner.py:619-658: one quartic per terminal speed (s(0), s'(0), s''(0) from the state; s'(T) = tv,
        s''(T) = 0), sampled on the cache's grid."""
        tc = time_cache
        a0, a1, a2 = frenet_state.s, frenet_state.s_d, frenet_state.s_dd / 2.0
        out = []
        for tv in np.asarray(target_velocities, dtype=float):
            a3, a4 = tc.quartic_A_inv @ np.array([tv - a1 - 2.0 * a2 * time, -2.0 * a2])
            out.append(LongitudinalProfile(
                t=tc.t, s=a0 + a1 * tc.t + a2 * tc.t2 + a3 * tc.t3 + a4 * tc.t4,
                s_d=a1 + 2.0 * a2 * tc.t + 3.0 * a3 * tc.t2 + 4.0 * a4 * tc.t3,
                s_dd=2.0 * a2 + 6.0 * a3 * tc.t + 12.0 * a4 * tc.t2, s_ddd=6.0 * a3 + 24.0 * a4 * tc.t))
        return out

    def _build_lateral_profiles(self, frenet_state, lateral_offsets, time: float, time_cache) -> list:
        """frenet_planner.py:660-701: one quintic per lateral target (d, d', d'' from the state; d(T) = di,
        d'(T) = d''(T) = 0)."""
        tc = time_cache
        a0, a1, a2 = frenet_state.d, frenet_state.d_d, frenet_state.d_dd / 2.0
        out = []
        for di in np.asarray(lateral_offsets, dtype=float):
            a3, a4, a5 = tc.quintic_A_inv @ np.array([di - a0 - a1 * time - a2 * time * time, -a1 - 2.0 * a2 * time,
                                                      -2.0 * a2])
            out.append(LateralProfile(
                d=a0 + a1 * tc.t + a2 * tc.t2 + a3 * tc.t3 + a4 * tc.t4 + a5 * tc.t5,
                d_d=a1 + 2.0 * a2 * tc.t + 3.0 * a3 * tc.t2 + 4.0 * a4 * tc.t3 + 5.0 * a5 * tc.t4,
                d_dd=2.0 * a2 + 6.0 * a3 * tc.t + 12.0 * a4 * tc.t2 + 20.0 * a5 * tc.t3,
                d_ddd=6.0 * a3 + 24.0 * a4 * tc.t + 60.0 * a5 * tc.t2))
        return out

    def _lattice_from(self, frenet_state, target_speed):
        fs = frenet_state
        req = PlanRequest(float(fs.s), float(fs.s_d), float(fs.s_dd), float(fs.d), float(fs.d_d),
                          target_speed=float(target_speed), last_kappa=float(fs.d_dd), is_frenet=True)
        self._engine.plan_batch([req])
        cost, _status, keep, _nt = self._engine.candidates(0)
        return cost, keep

    def _generate_frenet_paths(self, frenet_state, target_speed: float) -> List[FrenetPath]:
        """Every candidate of the lattice (grid, then the brake ladder) with its Frenet arrays and cost
        (frenet_planner.py:376-451); the Cartesian arrays are filled in by ``_calc_global_paths``."""
        cost, keep = self._lattice_from(frenet_state, target_speed)
        paths = []
        for i in range(len(cost)):
            full = self._engine.candidate_path(i)
            fp = FrenetPath(**{f: getattr(full, f) for f in self._FRENET_FIELDS})
            fp.cost = float(cost[i])
            fp.__dict__["_fot_global"] = ({f: getattr(full, f) for f in self._CART_FIELDS}, int(keep[i]))
            paths.append(fp)
        return paths

    def _generate_brake_candidates(self, frenet_state, target_speed: float) -> List[FrenetPath]:
        """The brake-ladder candidates alone (frenet_planner.py:453-503): the tail of the generation order; none
        below BRAKE_MIN_SPEED."""
        standing = FrenetState(frenet_state.s, 0.0, frenet_state.s_dd, frenet_state.d, frenet_state.d_d,
                               frenet_state.d_dd)
        n_grid = len(self._lattice_from(standing, target_speed)[0])          # (no ladder from a standing start)
        return self._generate_frenet_paths(frenet_state, target_speed)[n_grid:]

    def _calc_global_paths(self, fp_list: List[FrenetPath]) -> List[FrenetPath]:
        """Cartesian arrays of candidates made by ``_generate_frenet_paths``, cut to the valid prefix in lockstep
        with the Frenet arrays (frenet_planner.py:736-889: out-of-domain tail dropped, fewer than two samples or a
        singular point empty the path)."""
        for fp in fp_list:
            cart, keep = fp.__dict__["_fot_global"]
            for f in self._CART_FIELDS:
                setattr(fp, f, list(cart[f][:keep]))
            for f in self._FRENET_FIELDS:
                setattr(fp, f, list(getattr(fp, f)[:keep]))
        return fp_list

    def _path_collision_geometry(self, fp, dynamic_margin_inflation: float = 1.0):
        """(path_points, path_t, path_min, path_max, sq_rubicon, sq_rubicon_dyn) as the reference's tests read them
        (frenet_planner.py:1126-1179): one point per footprint circle and sample, the squared radii, the box.  A view
        for tests -- the kernels derive these points in registers."""
        n = min(len(fp.x), len(fp.t))
        if len(fp.x) == 0:
            return None
        pts = np.stack([np.asarray(fp.x[:n], float), np.asarray(fp.y[:n], float)], axis=1)
        t = np.asarray(fp.t[:n], float)
        radius = self.robot_radius
        if self.footprint is not None:
            radius = self.footprint.radius
            yaw = np.asarray(fp.yaw[:n], float)
            if len(yaw) < n:
                yaw = np.concatenate([yaw, np.full(n - len(yaw), yaw[-1] if len(yaw) else 0.0)])
            heading = np.stack([np.cos(yaw), np.sin(yaw)], axis=1)
            off = np.asarray(self.footprint.offsets, float)
            pts = (pts[None] + off[:, None, None] * heading[None]).reshape(-1, 2)
            t = np.tile(t, len(off))
        r_static = max(radius + self.obstacle_radius, 1e-6)
        r_dyn = r_static * dynamic_margin_inflation
        grow = max(r_static, r_dyn)
        return pts, t, pts.min(axis=0) - grow, pts.max(axis=0) + grow, r_static ** 2, r_dyn ** 2

    @staticmethod
    def _apply_stop_distance_filter(fp_dict: dict, max_stop_distance: float) -> None:
        """frenet_planner.py:307-324 on a categorised dict: 'ok' paths that do not come to rest within the travel
        move to 'stop_distance_error'."""
        ok, late = [], []
        for fp in fp_dict["ok"]:
            at_rest = len(fp.v) > 0 and abs(fp.v[-1]) <= 0.15                # STOP_SPEED_EPS
            travel = float(fp.s[-1] - fp.s[0]) if len(fp.s) > 0 else 0.0
            (ok if at_rest and travel <= max_stop_distance + 1e-6 else late).append(fp)
        fp_dict["ok"], fp_dict["stop_distance_error"] = ok, late

    @staticmethod
    def _select_best_path(path_dict: dict) -> Optional[FrenetPath]:
        """First minimum-cost entry of 'ok' (frenet_planner.py:1235-1259)."""
        best, best_cost = None, float("inf")
        for fp in path_dict["ok"]:
            if fp.cost < best_cost:
                best, best_cost = fp, fp.cost
        return best

    def candidate_table(self):
        """(cost, status, keep, n_t) per candidate of the last plan() call (diagnostic)."""
        return self._engine.candidates(0)

"""the obstacle-tensor producer in front of the planner, on the device.

Mirrors ``TrajectoryPredictor.process_prediction`` / ``predict_cv`` / the closest-to-mean pick of
``predict_single_best`` (reference src/prediction/trajectory_predictor.py:188-353) and the current-position
prepend of ``IntegratedSimulator._update_prediction`` (integrated_simulator.py:503-525).  The Social-GAN
forward passes themselves stay in PyTorch-ROCm; this turns their raw output into the planner's
``[S, P, T, 2]`` tensor without leaving the GPU.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _abi
from .planner import BatchPlanner

_dp = C.POINTER(C.c_double)


def _host_pd(a):
    return None if a is None else np.ascontiguousarray(a, dtype=np.float64).ctypes.data_as(_dp)


class PredictionResampler:
    def __init__(self, engine: BatchPlanner, pred_len: int = 12, sgan_dt: float = 0.4, sim_dt: float = 0.1,
                 plan_horizon: float = 5.0):
        self.engine = engine
        self.pred_len = int(pred_len)
        self.params = _abi.ResampleParams(float(sgan_dt), float(sim_dt), float(plan_horizon))
        self._lib = _abi.lib()

    @property
    def n_dense(self) -> int:
        return self._lib.fot_resample_n_dense(C.byref(self.params), self.pred_len)

    # -- host tensors in, host tensors out (drop-in for the reference methods) --------------------------
    def process_prediction(self, pred_traj: np.ndarray, anchor_pos: Optional[np.ndarray] = None,
                           staleness: float = 0.0, current: Optional[np.ndarray] = None,
                           want_sample_dist: bool = False):
        """pred_traj [pred_len, P, 2] -> [P, T, 2], or [S, pred_len, P, 2] -> [S, P, T, 2]."""
        pred = np.asarray(pred_traj)
        if pred.size == 0:
            return np.empty((0, 0, 2))
        single = pred.ndim == 3
        if single:
            pred = pred[None]
        if pred.ndim != 4 or pred.shape[-1] != 2:
            raise ValueError(f"Unexpected prediction shape: {np.shape(pred_traj)}")
        dt = np.float32 if pred.dtype == np.float32 else np.float64
        pred = np.ascontiguousarray(pred, dtype=dt)
        S, L, P = pred.shape[0], pred.shape[1], pred.shape[2]
        T = self._lib.fot_resample_n_dense(C.byref(self.params), L) + (0 if current is None else 1)
        out = np.zeros((S, P, T, 2), dtype=dt)
        dist = np.zeros(S) if want_sample_dist else None
        t_out = C.c_int32(0)
        code = _abi.F32 if dt == np.float32 else _abi.F64
        keep = (_host_pd(anchor_pos), _host_pd(current))
        _abi.check(self.engine._h, self._lib.fot_resample_predictions(
            self.engine._h, C.byref(self.params), S, L, P, pred.ctypes.data, code, keep[0], keep[1], float(staleness),
            out.ctypes.data, code, 0, C.byref(t_out), None if dist is None else dist.ctypes.data_as(_dp), None))
        assert t_out.value == T
        res = out[0] if single else out
        return (res, dist) if want_sample_dist else res

    def predict_cv(self, obs_traj: np.ndarray, staleness: float = 0.0, current: Optional[np.ndarray] = None,
                   float32_observations: bool = False):
        """obs_traj [obs_len, P, 2] (absolute) -> [P, T, 2]; velocity from the last two samples (:203-217).

        float32_observations: round the observations to float32 first and form the velocity in float32 -- what the
        reference does when the observer hands over its float tensors (observer.py:134)."""
        dt = np.float32 if float32_observations else np.float64
        obs = np.asarray(obs_traj).astype(dt)
        P = obs.shape[1]
        last = np.ascontiguousarray(obs[-1])
        prev = np.ascontiguousarray(obs[-2]) if obs.shape[0] >= 2 else None
        T = self.n_dense + (0 if current is None else 1)
        out = np.zeros((P, T, 2))
        t_out = C.c_int32(0)
        _abi.check(self.engine._h, self._lib.fot_predict_cv(
            self.engine._h, C.byref(self.params), self.pred_len, P, last.ctypes.data,
            None if prev is None else prev.ctypes.data, _abi.F32 if float32_observations else _abi.F64,
            _host_pd(current), float(staleness), out.ctypes.data, _abi.F64, 0, C.byref(t_out), None))
        return out

    @staticmethod
    def best_sample(sample_dist: np.ndarray) -> int:
        """np.argmin of the distances to the sample mean: first minimum (:349-350)."""
        return int(np.argmin(sample_dist))

    # -- device tensors (what the planner consumes through fot_plan_batch_device) ------------------------
    def resample_device(self, pred_ptr: int, pred_dtype, S: int, P: int, anchor_pos, current, staleness: float,
                        out_ptr: int, out_dtype, stream: Optional[int] = None, want_sample_dist: bool = False,
                        t_major: bool = False) -> Tuple[int, Optional[np.ndarray]]:
        """pred [S][pred_len][P][2] and out [S][P][T][2] are device pointers; returns (T, sample_dist).
        t_major: out is written [T][S][P][2] -- the layout the planner's broad phase reads fully coalesced; hand it on
        with ``PackedBatch(..., dyn_layout_tsp=True)`` / ``_abi.DYN_LAYOUT_TSP`` in ``dyn_dims``."""
        dist = np.zeros(S) if want_sample_dist else None
        t_out = C.c_int32(0)
        code = lambda d: _abi.F32 if np.dtype(d) == np.dtype(np.float32) else _abi.F64
        keep = (_host_pd(anchor_pos), _host_pd(current))
        _abi.check(self.engine._h, self._lib.fot_resample_predictions(
            self.engine._h, C.byref(self.params), S, self.pred_len, P, C.c_void_p(pred_ptr), code(pred_dtype), keep[0],
            keep[1], float(staleness), C.c_void_p(out_ptr), code(out_dtype),
            _abi.OUT_DEVICE | (_abi.OUT_TMAJOR if t_major else 0), C.byref(t_out),
            None if dist is None else dist.ctypes.data_as(_dp), C.c_void_p(stream) if stream else None))
        return t_out.value, dist

"""Handle teardown on the GPU: idempotent, bounded, never blocking on a caller's stream.

fot_destroy polls the handle's own streams and the event behind its last enqueue (hipStreamQuery / hipEventQuery) for a
bounded time instead of synchronising; what does not drain is left to the process teardown."""
import ctypes as C
import os
import time

import numpy as np
import pytest

from integrated_path_planning_amd import _abi, synthetic as syn
from integrated_path_planning_amd.batch import PackedBatch, request_from_instance
from integrated_path_planning_amd.planner import BatchPlanner

pytestmark = pytest.mark.gpu


def _planner():
    return BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=0, **syn.CONFIG3_PLANNER)


def _hip():
    """The one HIP runtime of this process (the copy libfot is bound to)."""
    _abi.lib()
    for name in (_abi.hip_soname(), "libamdhip64.so"):
        try:
            return C.CDLL(name, mode=getattr(os, "RTLD_NOLOAD", 4) | os.RTLD_NOW)
        except OSError:
            pass
    pytest.skip("HIP runtime handle not reachable through ctypes")


def test_close_is_idempotent_and_counted():
    lib = _abi.lib()
    n0 = lib.fot_live_handles()
    a, b = _planner(), _planner()
    assert lib.fot_live_handles() == n0 + 2
    raw = C.c_void_p(a._h.value)
    a.close()
    a.close()
    lib.fot_destroy(raw)                                           # a stale pointer: ignored, not a double free
    assert lib.fot_live_handles() == n0 + 1
    with _planner() as c:
        assert lib.fot_live_handles() == n0 + 2 and c._h
    assert lib.fot_live_handles() == n0 + 1
    b.close()
    assert lib.fot_live_handles() == n0


def test_destroy_after_the_callers_stream_is_gone():
    """The handle's last enqueue sat on a caller's stream that the caller has since destroyed (what PyTorch may do at
    exit): fot_destroy polls an EVENT, not the stream, and returns at once."""
    import torch
    hip = _hip()
    dev = torch.device("cuda", 0)
    bp = _planner()
    reqs = [request_from_instance(syn.config3_instance(s, S=4, P=8)) for s in range(3)]
    pb = PackedBatch(reqs, np.float32)
    want = bytes(bp.plan_packed(pb).records)
    dyn = torch.from_numpy(pb.dyn_xy).to(dev)
    out = torch.zeros(len(reqs) * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    st = C.c_void_p()
    assert hip.hipStreamCreate(C.byref(st)) == 0
    bp.plan_packed_device(pb.with_device_obstacles(None, dyn.data_ptr()), out.data_ptr(), st.value)
    hip.hipStreamSynchronize.argtypes = [C.c_void_p]
    hip.hipStreamDestroy.argtypes = [C.c_void_p]
    assert hip.hipStreamSynchronize(st) == 0
    assert out.cpu().numpy().tobytes() == want[: out.numel()]
    assert hip.hipStreamDestroy(st) == 0                            # the caller's stream is gone ...
    t0 = time.perf_counter()
    bp.close()                                                      # ... and the handle goes without it
    assert time.perf_counter() - t0 < 2.0


def test_destroy_does_not_wait_for_work_in_flight_beyond_its_budget(monkeypatch):
    """With a zero budget and a large batch still running on a caller's stream, fot_destroy returns immediately and
    leaves the buffers to the process (nothing is freed under the running kernels); the work itself completes."""
    import torch
    dev = torch.device("cuda", 0)
    monkeypatch.setenv("FOT_DESTROY_TIMEOUT_MS", "0")
    bp = _planner()
    reqs = [request_from_instance(syn.config3_instance(s)) for s in range(64)]
    pb = PackedBatch(reqs, np.float32)
    dyn = torch.from_numpy(pb.dyn_xy).to(dev)
    out = torch.zeros(len(reqs) * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream(dev)
    with torch.cuda.stream(st):
        for _ in range(8):
            bp.plan_packed_device(pb.with_device_obstacles(None, dyn.data_ptr()), out.data_ptr(), st.cuda_stream)
    t0 = time.perf_counter()
    bp.close()
    assert time.perf_counter() - t0 < 1.0
    st.synchronize()                                                # the enqueued calls still finish on leaked buffers
    rec = np.frombuffer(out.cpu().numpy().tobytes(), dtype=BatchPlanner.RESULT_DT)
    assert (rec["n_cand"] > 0).all()


def test_split_lanes_do_not_spin_on_record_flags(monkeypatch):
    """with FOT_LANES >= 2 a synchronous call the lanes split raised no record flags, and wait_records spun out
    its 20 ms before falling back to the stream.  Such calls are no longer armed: latency stays far below that."""
    monkeypatch.setenv("FOT_LANES", "2")
    bp2 = _planner()
    monkeypatch.delenv("FOT_LANES")
    bp1 = _planner()
    reqs = [request_from_instance(syn.config3_instance(s, S=4, P=8)) for s in range(48)]
    pb = PackedBatch(reqs, np.float32)
    want = bytes(bp1.plan_packed(pb).records)
    for _ in range(3):
        got = bp2.plan_packed(pb)
    ts = []
    for _ in range(10):
        t0 = time.perf_counter()
        got = bp2.plan_packed(pb)
        ts.append(time.perf_counter() - t0)
    assert bytes(got.records) == want
    assert np.median(ts) < 0.010, ts
    bp1.close()
    bp2.close()

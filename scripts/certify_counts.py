"""Diagnostic (GPU box): share of tiles the float32 certifying kernel gives up on, config 4 batches."""
import sys
import ctypes as C
import numpy as np
import torch
sys.path.insert(0, ".")
from integrated_path_planning_amd import _abi, synthetic as syn
from integrated_path_planning_amd.batch import PackedBatch, request_from_instance
from integrated_path_planning_amd.planner import BatchPlanner
bp = BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=0, **syn.CONFIG3_PLANNER)
bp.set_certify(2)
dev = torch.device("cuda", 0)
for b in range(3):
    reqs = [request_from_instance(syn.config3_instance(s)) for s in range(256 * b, 256 * (b + 1))]
    pb = PackedBatch(reqs, np.float32)
    dyn = torch.from_numpy(pb.dyn_xy).to(dev)
    out = torch.zeros(256 * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
    bp.plan_packed_device(pb.with_device_obstacles(None, dyn.data_ptr()), out.data_ptr(), None)
    bp.synchronize()
    t, r = C.c_int32(), C.c_int32()
    _abi.check(bp._h, bp._lib.fot_debug_certify_counts(bp._h, C.byref(t), C.byref(r)))
    print("batch", b, "tiles", t.value, "rest items", r.value, "share %.3f" % (r.value / max(t.value, 1)))
bp.close()

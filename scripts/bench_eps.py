"""Diagnostic: config 4 with a chance-constraint budget (eps > 0 keeps every candidate testing after its first hits)."""
import sys, time, os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import torch
from integrated_path_planning_amd.batch import request_from_instance
from integrated_path_planning_amd import _abi, synthetic as syn
from integrated_path_planning_amd.batch import PackedBatch
from integrated_path_planning_amd.planner import BatchPlanner

dev = torch.device("cuda", 0)
reqs = [request_from_instance(syn.config3_instance(s)) for s in range(256)]
pb = PackedBatch(reqs, obstacle_dtype=np.float32)
dyn_dev = torch.from_numpy(pb.dyn_xy).to(dev)
out_dev = torch.zeros(256 * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
bstruct = pb.with_device_obstacles(None, dyn_dev.data_ptr())
stream = torch.cuda.current_stream(dev)
for eps in (0.0, 0.1, 0.3):
    bp = BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=0, **dict(syn.CONFIG3_PLANNER, chance_epsilon=eps))
    for _ in range(5):
        bp.plan_packed_device(bstruct, out_dev.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    bp.profile(True); bp.profile_read(reset=True)
    t0 = time.perf_counter()
    for _ in range(30):
        bp.plan_packed_device(bstruct, out_dev.data_ptr(), stream.cuda_stream)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 30
    prof = bp.profile_read(reset=True)
    print(f"eps={eps}: {dt * 1e3:.3f} ms/step", {k: round(v['total_ms'] / max(v['launches'], 1), 4) for k, v in prof.items()})
    bp.close()

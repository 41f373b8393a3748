"""Diagnostic (GPU box): ONLY the headline loop of bench.py -- n plan calls in flight on n handles and streams, eight
config-4 batches rotated -- for a kernel trace without the other legs mixed in.
   overlap_loop.py [calls in flight = 3] [steps = 200]"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from integrated_path_planning_amd import _abi, synthetic as syn  # noqa: E402
from integrated_path_planning_amd.batch import PackedBatch, request_from_instance  # noqa: E402
from integrated_path_planning_amd.planner import BatchPlanner  # noqa: E402

n_ov = int(sys.argv[1]) if len(sys.argv) > 1 else 3
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
bstructs, keep = [], []
for b in range(8):
    pb = PackedBatch([request_from_instance(syn.config3_instance(s)) for s in range(256 * b, 256 * (b + 1))], np.float32)
    d = torch.from_numpy(pb.dyn_xy).to(dev)
    keep += [pb, d]
    bstructs.append(pb.with_device_obstacles(None, d.data_ptr()))
make = lambda: BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=0, **syn.CONFIG3_PLANNER)
leg = bench.Leg(n_ov, make, dev, bstructs, 256 * _abi.RESULT_BYTES, None, False, 256, 0)
elapsed, _, _ = leg.run(20, steps, False)
print("calls in flight %d: %.4f ms per step" % (n_ov, elapsed / steps * 1e3))
leg.close()

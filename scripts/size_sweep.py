"""Diagnostic (GPU box): k_evaluate time of a config-3 batch of n egos, walked in one piece (the batch kernels) --
how the launch time grows with the batch: steps mean rounds of resident workgroups, a line means throughput."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from integrated_path_planning_amd import synthetic as syn                                   # noqa: E402
from integrated_path_planning_amd.batch import PackedBatch, request_from_instance          # noqa: E402
from integrated_path_planning_amd.planner import BatchPlanner                               # noqa: E402

bp = BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=0, **syn.CONFIG3_PLANNER)
bp.set_eval_segments(int(os.environ.get("FOT_SWEEP_SEGMENTS", "1")))
reqs = [request_from_instance(syn.config3_instance(s)) for s in range(512)]
for n in (32, 64, 96, 112, 128, 144, 160, 192, 224, 256, 288, 320, 384, 448, 512):
    pbs = [PackedBatch(reqs[r:r + n] if r + n <= 512 else reqs[:n], np.float32) for r in (0, 64)]
    for b in pbs:
        bp.plan_packed(b)
    bp.profile(True); bp.profile_read(reset=True)
    for it in range(40):
        bp.plan_packed(pbs[it % 2])
    pr = bp.profile_read(reset=True)
    bp.profile(False)
    ev = pr["k_evaluate"]["total_ms"] / pr["k_evaluate"]["launches"] * 1e3
    print("n_inst %3d  groups/queue %4d  k_evaluate %6.1f us  (%.3f us per instance)" % (n, (n + 7) // 8 * 9, ev, ev / n), flush=True)

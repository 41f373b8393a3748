"""Diagnostic (GPU box): k_evaluate time of a config-3 batch of n egos with the time range cut into 1 / 2 / 4 segments
(fot_debug_set_eval_segments) -- where the automatic choice of launch_evaluate should switch."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from integrated_path_planning_amd import synthetic as syn                                   # noqa: E402
from integrated_path_planning_amd.batch import PackedBatch, request_from_instance          # noqa: E402
from integrated_path_planning_amd.planner import BatchPlanner                               # noqa: E402

bp = BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=0, **syn.CONFIG3_PLANNER)
for n in [int(v) for v in os.environ.get("FOT_SWEEP_N", "1 2 4 6 8 12 16 24 32 48 64").split()]:
    pbs = [PackedBatch([request_from_instance(syn.config3_instance(100 * r + s)) for s in range(n)], np.float32)
           for r in range(4)]
    row = []
    for n_seg in (1, 2, 3, 4):
        bp.set_eval_segments(n_seg)
        for b in pbs:
            bp.plan_packed(b)
        bp.profile(True); bp.profile_read(reset=True)
        for it in range(60):
            bp.plan_packed(pbs[it % 4])
        pr = bp.profile_read(reset=True)
        bp.profile(False)
        row.append(pr["k_evaluate"]["total_ms"] / pr["k_evaluate"]["launches"] * 1e3)
    print("n_inst %3d  tiles %5d  k_evaluate us: " % (n, 36 * n) + "  ".join("seg%d %6.1f" % (i + 1, v) for i, v in enumerate(row)), flush=True)

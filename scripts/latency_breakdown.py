"""Diagnostic (GPU box): per-kernel device time of a ONE-ego plan call (config 2 and config 3) next to its wall time."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from integrated_path_planning_amd import synthetic as syn                                   # noqa: E402
from integrated_path_planning_amd.batch import PackedBatch, request_from_instance          # noqa: E402
from integrated_path_planning_amd.planner import BatchPlanner                               # noqa: E402

for name, pk, mk in (("config2", syn.CONFIG2_PLANNER, syn.config2_instance), ("config3", syn.CONFIG3_PLANNER, syn.config3_instance)):
    bp = BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=0, **pk)
    pk8 = [PackedBatch([request_from_instance(mk(s))], np.float32) for s in range(8)]
    for b in pk8:
        bp.plan_packed(b)
    ts = []
    for it in range(500):
        t1 = time.perf_counter(); bp.plan_packed(pk8[it % 8]); ts.append(time.perf_counter() - t1)
    bp.profile(True); bp.profile_read(reset=True)
    for it in range(200):
        bp.plan_packed(pk8[it % 8])
    pr = bp.profile_read(reset=True)
    print(name, "wall p50 %.1f us" % (np.percentile(ts, 50) * 1e6),
          {k: round(v["total_ms"] / max(v["launches"], 1) * 1e3, 1) for k, v in pr.items()}, "(us per kernel, HIP events)")

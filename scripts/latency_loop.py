"""One-ego plan calls in a loop (config 2, then config 3): the workload of scripts/latency_gaps.sh.  Prints the wall
p50 per call (inflated when a profiler is attached; run it bare for the true figure) and, as a floor, the round trip of
the smallest synchronous call the library has (fot_frenet_state_batch of one ego: one launch + one wait)."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from integrated_path_planning_amd import synthetic as syn                                   # noqa: E402
from integrated_path_planning_amd.batch import PackedBatch, request_from_instance          # noqa: E402
from integrated_path_planning_amd.planner import BatchPlanner                               # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 400
for name, pk, mk in (("config2", syn.CONFIG2_PLANNER, syn.config2_instance), ("config3", syn.CONFIG3_PLANNER, syn.config3_instance)):
    bp = BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=0, **pk)
    pk8 = [PackedBatch([request_from_instance(mk(s))], np.float32) for s in range(8)]
    for _ in range(3):
        for b in pk8:
            bp.plan_packed(b)
    ts = []
    for it in range(N):
        t1 = time.perf_counter(); bp.plan_packed(pk8[it % 8]); ts.append(time.perf_counter() - t1)
    e = request_from_instance(mk(0))
    x, y, yaw, v, a = (np.array([getattr(e, f)]) for f in ("x", "y", "yaw", "v", "a"))
    tf = []
    for it in range(N):
        t1 = time.perf_counter(); bp.nearest_s_arrays(x, y, yaw, v, a, np.array([np.nan])); tf.append(time.perf_counter() - t1)
    print("%s: plan call wall p50 %.1f us (p95 %.1f); one launch + one wait (nearest point of one ego) p50 %.1f us"
          % (name, np.percentile(ts, 50) * 1e6, np.percentile(ts, 95) * 1e6, np.percentile(tf, 50) * 1e6))
    bp.close()

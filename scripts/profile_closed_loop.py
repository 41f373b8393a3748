"""Host-side time split of the batched closed-loop driver: lock steps of 64 scenario_01 episodes, fused (two libfot
calls per step) and unfused (five).  First the plain wall time per step, then the libfot calls timed by wrapping the
engine's methods (perf_counter around each; cProfile's own overhead distorts a sub-millisecond step -- see
cprofile_closed_loop.py for where Python spends its share); what is left of the step is NumPy / Python."""
import json
import os
import sys
import time
from collections import defaultdict

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from integrated_path_planning_amd.closed_loop import BatchedClosedLoop  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "closed_loop", "reference_cv_episodes.npz"), allow_pickle=False)
cfg = json.loads(str(z["meta"]))["config"]
N = 200


def run(fused, wrapped):
    loop = BatchedClosedLoop(cfg, [z["base_ped_traj"]] * 64, fused=fused)
    spent, calls = defaultdict(float), defaultdict(int)

    def wrap(obj, name):
        f = getattr(obj, name)

        def g(*a, **k):
            t0 = time.perf_counter()
            try:
                return f(*a, **k)
            finally:
                spent[name] += time.perf_counter() - t0
                calls[name] += 1
        setattr(obj, name, g)

    if wrapped:
        for nm in ("plan_arrays", "safety_metrics_cat", "nearest_s_arrays", "loop_plan", "loop_observe"):
            wrap(loop.engine, nm)
        wrap(loop.resampler, "predict_cv")
    for _ in range(20):
        loop.step()
    spent.clear(); calls.clear()
    t0 = time.perf_counter()
    for _ in range(N):
        loop.step()
    wall = time.perf_counter() - t0
    print("%s, %s: %.4f ms per lock step (%d episodes running)" % ("fused" if fused else "unfused",
          "calls timed" if wrapped else "plain", wall / N * 1e3, int(loop.alive.sum())))
    lib = 0.0
    for k, v in sorted(spent.items(), key=lambda kv: -kv[1]):
        print("  %-22s %.4f ms/step  (%.2f calls/step)" % (k, v / N * 1e3, calls[k] / N))
        lib += v
    if wrapped:
        print("  %-22s %.4f ms/step" % ("python + numpy", (wall - lib) / N * 1e3))
    loop.close()


for fused in (True, False):
    run(fused, False)
    run(fused, True)

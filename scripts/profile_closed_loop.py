"""Host-side profile (cProfile) of the batched closed-loop driver: 100 lock steps of 64 scenario_01 episodes."""
import cProfile
import json
import os
import pstats
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from integrated_path_planning_amd.closed_loop import BatchedClosedLoop  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "closed_loop", "reference_cv_episodes.npz"), allow_pickle=False)
cfg = json.loads(str(z["meta"]))["config"]
loop = BatchedClosedLoop(cfg, [z["base_ped_traj"]] * 64)
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for _ in range(100):
    loop.step()
wall = time.perf_counter() - t0
pr.disable()
print("ms per lock step", wall / 100 * 1e3)
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

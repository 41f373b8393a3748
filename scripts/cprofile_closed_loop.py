"""cProfile of the batched closed-loop driver's lock step (distorts the absolute time; shows where Python spends it)."""
import cProfile
import json
import os
import pstats
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401,E402
from integrated_path_planning_amd.closed_loop import BatchedClosedLoop  # noqa: E402

z = np.load(os.path.join(ROOT, "tests", "golden", "closed_loop", "reference_cv_episodes.npz"), allow_pickle=False)
cfg = json.loads(str(z["meta"]))["config"]
loop = BatchedClosedLoop(cfg, [z["base_ped_traj"]] * 64)
for _ in range(20):
    loop.step()
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    loop.step()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)

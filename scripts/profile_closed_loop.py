import json, time, sys
import numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import torch
from integrated_path_planning_amd.closed_loop import BatchedClosedLoop
import integrated_path_planning_amd.closed_loop as cl
z = np.load('/root/repo/tests/golden/closed_loop/reference_cv_episodes.npz', allow_pickle=False)
cfg = json.loads(str(z['meta']))['variants']['base']['config']
loop = BatchedClosedLoop(cfg, [z['base_ped_traj']] * 64)
import cProfile, pstats
pr = cProfile.Profile()
pr.enable()
t0 = time.perf_counter()
for _ in range(100):
    loop.step()
wall = time.perf_counter() - t0
pr.disable()
print('ms per lock step', wall / 100 * 1e3)
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)

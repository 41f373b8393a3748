"""Diagnostic (GPU box): where a native lock step (fot_loop_step) of 64 scenario_01 episodes spends its time --
the one libfot call, the frame preparation, the history bookkeeping."""
import json, os, sys, time
from collections import defaultdict
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa
from integrated_path_planning_amd.closed_loop import BatchedClosedLoop
z = np.load(os.path.join(ROOT, "tests", "golden", "closed_loop", "reference_cv_episodes.npz"), allow_pickle=False)
cfg = json.loads(str(z["meta"]))["config"]
n_ep = int(sys.argv[1]) if len(sys.argv) > 1 else 64
loop = BatchedClosedLoop(cfg, [z["base_ped_traj"]] * n_ep)
spent, calls = defaultdict(float), defaultdict(int)
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter()
        try: return f(*a, **k)
        finally:
            spent[name] += time.perf_counter() - t0; calls[name] += 1
    setattr(obj, name, g)
for nm in ("loop_step", "gather_paths", "_loop_frame"):
    wrap(loop.engine if nm != "_loop_frame" else loop, nm)
wrap(loop, "_loop_frame"); wrap(loop, "_advance_pedestrians"); wrap(loop, "_step_native")
for _ in range(20): loop.step()
spent.clear(); calls.clear()
t0 = time.perf_counter(); n = 0
for _ in range(200):
    if loop.step() == 0: break
    n += 1
wall = time.perf_counter() - t0
print("episodes %d: %.4f ms per lock step" % (n_ep, wall / n * 1e3))
for k, v in sorted(spent.items(), key=lambda kv: -kv[1]):
    print("  %-22s %.4f ms per step (%d calls)" % (k, v / n * 1e3, calls[k]))
loop.close()

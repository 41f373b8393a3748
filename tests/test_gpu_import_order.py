"""One HIP runtime per process, whatever the import order, and a deterministic exit.

PyTorch-ROCm bundles its own HIP/HSA runtime; libfot asks for the system SONAME.  `_abi.lib()` binds libfot to the
copy torch will use, so a planner created BEFORE `import torch` and a torch device tensor allocated afterwards share
one runtime (north_star: Social-GAN on PyTorch-ROCm feeds the kernel in the same process).  Runs in a child process:
the test session itself may already have imported torch.

The child deliberately does NOT close its planner and keeps a torch stream and tensors alive as module globals -- the
state of the one child that never came back in round 3 (gpurun_out/r3_t29.log).  Teardown is now `_abi`'s atexit hook
(every open planner closed before module teardown; `fot_destroy` polls, never blocks on a caller's stream): the child
must print the three phase markers and exit 0 within the time limit.  No retry: a child that hangs fails the test with
the phase it reached and its faulthandler stack."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import atexit, faulthandler, re, sys
faulthandler.dump_traceback_later(120, exit=True)              # a hang shows where (all threads), and ends the child
def _after_hook():                                             # registered first = runs LAST, behind _abi's exit hook
    from integrated_path_planning_amd import _abi
    print("PHASE planners-closed live=%d" % _abi.lib().fot_live_handles(), flush=True)
atexit.register(_after_hook)
import numpy as np
assert "torch" not in sys.modules
from integrated_path_planning_amd import _abi, synthetic as syn
from integrated_path_planning_amd.batch import PackedBatch, request_from_instance
from integrated_path_planning_amd.planner import BatchPlanner
bp = BatchPlanner(waypoints=(syn.STRAIGHT_WX, syn.STRAIGHT_WY), device=0, **syn.CONFIG3_PLANNER)   # HIP initialised by libfot
reqs = [request_from_instance(syn.config3_instance(s, S=4, P=8)) for s in range(3)]
pb = PackedBatch(reqs, np.float32)
host = bp.plan_packed(pb)
assert "torch" not in sys.modules
import torch                                                   # ... and only now torch
dev = torch.device("cuda", 0)
dyn = torch.from_numpy(pb.dyn_xy).to(dev)                      # torch device memory
out = torch.zeros(len(reqs) * _abi.RESULT_BYTES, dtype=torch.uint8, device=dev)
st = torch.cuda.Stream(dev)                                    # a caller's stream: the handle's last enqueue sits on it
with torch.cuda.stream(st):
    bp.plan_packed_device(pb.with_device_obstacles(None, dyn.data_ptr()), out.data_ptr(), st.cuda_stream)
st.synchronize()
assert out.cpu().numpy().tobytes() == bytes(host.records)[: out.numel()]
maps = open("/proc/self/maps").read()
hip = sorted(set(re.findall(r"/\S*libamdhip64\S*", maps)))
hsa = sorted(set(re.findall(r"/\S*libhsa-runtime64\S*", maps)))
assert len(hip) == 1 and len(hsa) == 1, (hip, hsa)
print("ONE_RUNTIME_OK", _abi.hip_runtime_path, flush=True)
assert _abi.lib().fot_live_handles() == 1
atexit.register(lambda: print("PHASE atexit-begin", flush=True))   # registered last = runs first
print("PHASE script-end", flush=True)                          # bp, st, dyn, out stay alive as module globals
"""


def test_planner_before_torch_shares_one_hip_runtime_and_the_process_exits():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    try:
        p = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    except subprocess.TimeoutExpired as e:                         # (past faulthandler's exit: stuck after finalisation)
        out = (e.stdout or b"").decode(errors="replace") if isinstance(e.stdout, bytes) else (e.stdout or "")
        err = (e.stderr or b"").decode(errors="replace") if isinstance(e.stderr, bytes) else (e.stderr or "")
        pytest.fail("the child did not exit; phases reached:\n" + out[-2000:] + "\nstderr:\n" + err[-4000:])
    assert p.returncode == 0, "phases reached:\n" + p.stdout[-2000:] + "\nstderr:\n" + p.stderr[-4000:]
    assert "ONE_RUNTIME_OK" in p.stdout
    assert "PHASE script-end" in p.stdout
    assert "PHASE planners-closed live=0" in p.stdout, p.stdout     # the hook closed the planner the script left open

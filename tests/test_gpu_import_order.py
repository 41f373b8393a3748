"""One HIP runtime per process, whatever the import order.

PyTorch-ROCm bundles its own HIP/HSA runtime; libfot asks for the system SONAME.  `_abi.lib()` binds libfot to the
copy torch will use, so a planner created BEFORE `import torch` and a torch device tensor allocated afterwards share
one runtime (north_star: Social-GAN on PyTorch-ROCm feeds the kernel in the same process).  Runs in a child process:
the test session itself may already have imported torch."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import faulthandler, re, sys
faulthandler.dump_traceback_later(150, exit=True)              # a hang shows where, and ends the child
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
st = torch.cuda.current_stream(dev)
bp.plan_packed_device(pb.with_device_obstacles(None, dyn.data_ptr()), out.data_ptr(), st.cuda_stream)
torch.cuda.synchronize(dev)
assert out.cpu().numpy().tobytes() == bytes(host.records)[: out.numel()]
maps = open("/proc/self/maps").read()
hip = sorted(set(re.findall(r"/\S*libamdhip64\S*", maps)))
hsa = sorted(set(re.findall(r"/\S*libhsa-runtime64\S*", maps)))
assert len(hip) == 1 and len(hsa) == 1, (hip, hsa)
print("ONE_RUNTIME_OK", _abi.hip_runtime_path, flush=True)
bp.close()
"""


def test_planner_before_torch_shares_one_hip_runtime():
    env = dict(os.environ, PYTHONPATH=ROOT + os.pathsep + os.environ.get("PYTHONPATH", ""))
    for attempt in range(2):
        p = subprocess.run([sys.executable, "-c", CHILD], cwd=ROOT, env=env, capture_output=True, text=True, timeout=400)
        # A second process on the card while this one holds its handles and a torch context: once in some fifty runs of
        # the whole suite on the shared pool the child did not come back (in isolation: never, tens of runs).  The child
        # ends itself after 150 s with its Python stack; one more attempt then, and the stack is shown either way.
        if p.returncode == 0 or "Timeout" not in p.stderr:
            break
        print("child timed out, stack:\n" + p.stderr[-3000:])
    assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
    assert "ONE_RUNTIME_OK" in p.stdout

"""Build-time guard for the hand-issued scalar loads of the evaluation kernels.

FusedSink (csrc/fot_kernels.hip) issues `s_load_dwordx16` / `s_load_dword` in one inline-asm statement and waits for them
in a later one, so that the chunk being tested and the next one overlap.  The compiler does not know that the destination
registers are in flight in between; if its register allocator parks or reuses one of them there (it does under scalar
register pressure -- a kernel variant with 70 spilled SGPRs did, and faulted on the GPU), the data lands in whatever lives
in those registers by then.  scripts/isa_check_async.py looks for exactly that in the gfx950 ISA of the current sources."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not (os.path.exists("/opt/rocm/bin/hipcc") or shutil.which("hipcc")), reason="hipcc not available")
def test_no_allocator_moves_on_registers_in_flight():
    subprocess.run([os.path.join(ROOT, "scripts", "isa.sh")], check=True, stdout=subprocess.DEVNULL, timeout=600)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "isa_check_async.py"), "/tmp/isa/fot.s"],
                       capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("hand-issued scalar loads, 0 instructions") >= 3, r.stdout      # all three evaluation kernels seen

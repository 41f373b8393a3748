"""The build-time guard for the hand-issued scalar loads of the evaluation kernels (scripts/isa_check_async.py).

csrc/Makefile runs the guard on the device ISA of EVERY build and fails the build on a finding; here: that the guard
itself catches what it is there for (synthetic ISA with each kind of hazard), and that the ISA of the current sources
passes it under the flag sets the diagnostic scripts build (-DFOT_TIMELINE)."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GUARD = os.path.join(ROOT, "scripts", "isa_check_async.py")
HIPCC = "/opt/rocm/bin/hipcc" if os.path.exists("/opt/rocm/bin/hipcc") else shutil.which("hipcc")

META = """
  - .name:           _ZN3fot{n}{name}EPKv
    .private_segment_fixed_size: {scratch}
    .sgpr_count:     100
    .sgpr_spill_count: 0
    .vgpr_count:     100
    .vgpr_spill_count: {vspill}
    .wavefront_size: 64
"""


def synthetic(loop_body, scratch=0, vspill=0, callee="\ts_mov_b32 s4, 0\n\ts_setpc_b64 s[30:31]"):
    """three evaluation kernels with `loop_body` between a hand-issued load of s[16:31] and its wait, and one callee"""
    out = ["\t.text"]
    for name in ("k_evaluate", "k_evaluate_split", "k_evaluate_group"):
        sym = f"_ZN3fot{len(name)}{name}EPKv"
        out += [f"{sym}:                        ; @{sym}", "; %bb.0:", "\ts_load_dwordx2 s[60:61], s[0:1], 0x0",
                "\ts_waitcnt lgkmcnt(0)", ".LBB0_1:", "\t;;#ASMSTART", "\ts_load_dwordx16 s[16:31], s[60:61], 0",
                "\t;;#ASMEND", loop_body, "\t;;#ASMSTART", "\ts_waitcnt lgkmcnt(0)", "\t;;#ASMEND",
                "\tv_pk_add_f32 v[0:1], v[2:3], s[16:17]", "\ts_cbranch_scc1 .LBB0_1", "\ts_endpgm", ".Lfunc_end0:"]
    out += ["_ZN3fot17yaw_step_over_capEdddd:         ; @f", callee, ".Lfunc_end9:"]
    out += ["amdhsa.kernels:"]
    for name in ("k_evaluate", "k_evaluate_split", "k_evaluate_group"):
        out.append(META.format(n=len(name), name=name, scratch=scratch, vspill=vspill))
    return "\n".join(out) + "\n"


def run_guard(text, tmp_path, *args):
    p = tmp_path / "k.s"
    p.write_text(text)
    extra = [] if "--handoff" in args else ["--no-handoff-check"]     # (the hazard cases below hold no selection code)
    args = [a for a in args if a != "--handoff"]
    r = subprocess.run([sys.executable, GUARD, str(p), "--min-kernels", "3", *extra, *args], capture_output=True, text=True, timeout=60)
    return r.returncode, r.stdout + r.stderr


def test_guard_passes_clean_code(tmp_path):
    rc, out = run_guard(synthetic("\tv_pk_mul_f32 v[4:5], v[4:5], v[4:5]\n\ts_add_i32 s40, s40, 1"), tmp_path)
    assert rc == 0, out
    assert out.count("0 instructions touching") == 3


@pytest.mark.parametrize("hazard", [
    "\tv_writelane_b32 v127, s20, 3",                      # register allocator parks a destination in a VGPR lane
    "\ts_mov_b64 s[18:19], s[40:41]",                      # ... reuses it
    "\ts_add_i32 s31, s31, 1",                             # any scalar instruction that writes it
    "\tv_pk_add_f32 v[0:1], v[2:3], s[22:23]",             # a read before the wait
    "\ts_cbranch_scc0 .LBB0_9\n\ts_nop 0\n.LBB0_9:\n\ts_and_b64 s[16:17], s[16:17], exec",   # past a branch and a label
    "\ts_swappc_b64 s[30:31], s[34:35]",                   # a call whose callee names an SGPR in flight
])
def test_guard_catches_hazards(tmp_path, hazard):
    callee = "\ts_mov_b32 s17, 0\n\ts_setpc_b64 s[30:31]" if "swappc" in hazard else "\ts_setpc_b64 s[30:31]"
    rc, out = run_guard(synthetic(hazard, callee=callee), tmp_path)
    assert rc == 1, out
    assert "in flight" in out


def test_guard_catches_missing_early_clobber(tmp_path):
    body = "\t;;#ASMSTART\n\ts_load_dword s60, s[60:61], 0x0\n\ts_load_dword s60, s[60:61], 0x40\n\t;;#ASMEND"
    rc, out = run_guard(synthetic(body), tmp_path)
    assert rc == 1 and "overlaps the address" in out, out


def test_guard_rejects_scratch_and_vector_spills(tmp_path):
    clean = "\ts_nop 0"
    assert run_guard(synthetic(clean, scratch=104, vspill=29), tmp_path)[0] == 1
    assert run_guard(synthetic(clean, scratch=104, vspill=29), tmp_path, "--allow-scratch")[0] == 0
    assert run_guard(synthetic(clean), tmp_path)[0] == 0


def handoff(store_flag=" sc1", wait="\ts_waitcnt vmcnt(0)", load_flag=" sc1", loads_first=False, n_stores=11):
    """the selection hand-off as the evaluation kernels hold it: partial result out, wait, count, wait, partials in"""
    st = "\n".join(f"\tglobal_store_dword v[2:3], v{4 + i}, off offset:{4 * i}{store_flag}" for i in range(n_stores))
    ld = "\n".join(f"\tglobal_load_dword v{20 + i}, v[0:1], off offset:{4 * i}{load_flag}" for i in range(11))
    tail = (ld + "\n\ts_waitcnt vmcnt(0)") if loads_first else ("\ts_waitcnt vmcnt(0)\n" + ld)
    return st + "\n" + wait + "\n\tglobal_atomic_add v1, v1, v2, s[6:7] sc0\n" + tail


def test_guard_checks_the_fence_free_selection(tmp_path):
    """the selection folded into the evaluation has no release / acquire pair; it is correct only while the
    ISA keeps sc1 on the partial-result stores and loads, a wait between stores and count, and the loads behind the wait
    that returns the count.  The guard asserts exactly that on every build."""
    clean = "\tv_pk_mul_f32 v[4:5], v[4:5], v[4:5]\n"
    assert run_guard(synthetic(clean + handoff()), tmp_path, "--handoff")[0] == 0
    for bad, word in ((handoff(store_flag=""), "store without sc1"),
                      (handoff(wait="\ts_nop 0"), "no s_waitcnt vmcnt(0) between"),
                      (handoff(load_flag=""), "agent-coherent loads behind"),
                      (handoff(loads_first=True), "before the tile counter has returned"),
                      (handoff(n_stores=9), "only 9 stores"),
                      ("\ts_nop 0", "no global_atomic_add")):
        rc, out = run_guard(synthetic(clean + bad), tmp_path, "--handoff")
        assert rc == 1 and word in out, (word, out)


def test_makefile_runs_the_guard():
    mk = open(os.path.join(ROOT, "integrated_path_planning_amd", "csrc", "Makefile")).read()
    assert "isa_check_async.py" in mk and "-save-temps=obj" in mk and "exit 1" in mk


@pytest.mark.skipif(not HIPCC, reason="hipcc not available")
@pytest.mark.parametrize("flags", [[], ["-DFOT_TIMELINE"]], ids=["default", "timeline"])
def test_current_sources_pass_under_every_flag_set_the_scripts_build(tmp_path, flags):
    s = tmp_path / "fot.s"
    subprocess.run([HIPCC, "-O3", "-std=c++17", "-ffp-contract=on", "--offload-arch=gfx950", "-Wno-unused-function", "-S", "--cuda-device-only",
                    *flags, "-o", str(s), os.path.join(ROOT, "integrated_path_planning_amd", "csrc", "fot_kernels.hip")],
                   check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
    r = subprocess.run([sys.executable, GUARD, str(s)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert r.stdout.count("hand-issued scalar loads, 0 instructions") == 3, r.stdout      # every evaluation kernel seen

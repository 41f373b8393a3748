"""bench.py's launch contract, checked without a GPU: `--gpus N` with no rank environment must start N ranks as
child processes of `torch.distributed.run` (never re-exec a process that touched the GPU)."""
import importlib.util
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _load_bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_self_launch_command(monkeypatch):
    bench = _load_bench()
    seen = {}

    def fake_run(cmd, env=None, **kw):
        seen["cmd"], seen["env"] = cmd, env
        return types.SimpleNamespace(returncode=7)

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "5", "--warmup", "1"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    args = bench.parse_args()
    assert bench.self_launch(args) == 7                      # the children's exit code is passed through
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 0 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "5", "--warmup", "1"]
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"


def test_main_self_launches_before_touching_torch(monkeypatch):
    """With --gpus 2 and no WORLD_SIZE, main() must hand over to the children and exit with their code without
    importing torch.cuda state itself (torch may be imported by the test session; what matters is the early exit)."""
    bench = _load_bench()
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(bench, "self_launch", lambda a: 0)
    try:
        bench.main()
    except SystemExit as e:
        assert e.code == 0
    else:
        raise AssertionError("main() continued in the parent process")


def test_source_hash_is_shared_with_the_profile_summary():
    """profiles/pmc.json is only trusted for the kernel sources it was taken on: both sides hash the same files."""
    bench = _load_bench()
    spec = importlib.util.spec_from_file_location("pmc_summary", os.path.join(ROOT, "scripts", "pmc_summary.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.KERNEL_SOURCES == bench.KERNEL_SOURCES
    assert mod.kernel_source_hash() == bench.kernel_source_hash()

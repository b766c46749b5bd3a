"""CPU: `python bench.py --gpus N` typed without a launcher starts its own ranks -- the command it builds is the one the driver uses for
N > 1 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py <same args>`), as a
child process (subprocess, never os.exec*), before torch is imported; under a launcher (WORLD_SIZE set) nothing is started."""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_launch_ranks_command(monkeypatch):
    b = _bench()
    seen = {}

    def fake_call(cmd, env=None):
        seen["cmd"], seen["env"] = cmd, env
        return 7

    import subprocess
    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "3", "--warmup", "1"])
    monkeypatch.delenv("MASTER_PORT", raising=False)
    assert b.launch_ranks(4) == 7  # the child's exit code is the launcher's
    cmd = seen["cmd"]
    assert cmd[0] == sys.executable and cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "4" and cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    assert 1024 < int(cmd[cmd.index("--master-port") + 1]) < 65536
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "3", "--warmup", "1"]  # the same arguments, after the script
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"  # dmabuf IPC: RCCL across processes needs it on this platform


def test_main_starts_ranks_only_without_a_launcher(monkeypatch):
    b = _bench()
    calls = []
    monkeypatch.setattr(b, "launch_ranks", lambda n: calls.append(n) or 0)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2"])
    try:
        b.main()
    except SystemExit as e:
        assert e.code == 0
    assert calls == [2]
    # under a launcher: a rank, no child -- it gets as far as the GPU check (this container has none) or the world-size check
    calls.clear()
    monkeypatch.setenv("WORLD_SIZE", "2")
    monkeypatch.setenv("RANK", "0")
    try:
        b.main()
        raised = None
    except SystemExit as e:
        raised = e
    except Exception as e:  # noqa: BLE001 -- no GPU here: anything but a launch
        raised = e
    assert calls == [] and raised is not None

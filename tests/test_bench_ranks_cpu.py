"""CPU-only: `bench.py --gpus N` starts N rank processes itself when no launcher set WORLD_SIZE, refuses to run more ranks than
there are GPUs, and gives every rank the torch.distributed environment the contract names."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_more_ranks_than_gpus_fails_loudly():
    import torch
    have = torch.cuda.device_count()
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(have + 2), "--steps", "1", "--warmup", "0"],
                       capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "GPU(s) visible" in r.stderr and r.stdout.strip() == ""


def test_spawn_sets_rank_environment(monkeypatch):
    import bench
    import torch
    started = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None):
            started.append((cmd, env, stdout))
            self.rank = int(env["RANK"])

        def poll(self):
            return 0

        def terminate(self):
            pass

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 4)
    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "4", "--steps", "2"])
    assert bench.spawn_ranks(bench.parse(["--gpus", "4", "--steps", "2"])) == 0
    assert len(started) == 4
    ports = set()
    for r, (cmd, env, stdout) in enumerate(started):
        assert cmd[1].endswith("bench.py") and cmd[2:] == ["--gpus", "4", "--steps", "2"]
        assert env["RANK"] == env["LOCAL_RANK"] == str(r) and env["WORLD_SIZE"] == "4" and env["MASTER_ADDR"] == "127.0.0.1"
        assert env["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
        ports.add(env["MASTER_PORT"])
        assert (stdout is None) == (r == 0)  # only rank 0 owns the command's stdout (the ONE JSON line)
    assert len(ports) == 1


def test_failed_rank_fails_the_run(monkeypatch):
    import bench
    import torch
    killed = []

    class FakeProc:
        def __init__(self, cmd, env=None, stdout=None):
            self.rank = int(env["RANK"])

        def poll(self):
            return 3 if self.rank == 1 else None if self.rank == 0 and not killed else 0

        def terminate(self):
            killed.append(self.rank)

    monkeypatch.setattr(torch.cuda, "device_count", lambda: 2)
    monkeypatch.setattr(bench.subprocess, "Popen", FakeProc)
    assert bench.spawn_ranks(bench.parse(["--gpus", "2"])) == 3
    assert killed == [0]

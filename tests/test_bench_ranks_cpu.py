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


def test_also_runs_are_valid_bench_invocations_and_collect_one_line_each(monkeypatch):
    """r6: the `also` object of the default line -- every side run is an argument list this script's own parser accepts (a typo there
    would cost the driver's line its Modular / 8K / EPF figures), never one that would recurse into `also`, and also_runs() keeps the
    child's value / ms / frac / traffic / cpu_baseline or an `error`, whatever the child does"""
    import json
    import bench
    keys = [k for k, _, _ in bench.ALSO_RUNS]
    assert keys == ["modular8k", "modular8k_x4", "modular1080p", "vardct8k_pq", "vardct4k_epf1", "vardct4k_epf3"]
    calls = []

    class R:
        def __init__(self, rc, out):
            self.returncode, self.stdout, self.stderr = rc, out, "boom"

    def fake_run(cmd, env=None, capture_output=None, text=None, timeout=None):
        a = bench.parse(cmd[2:])  # argparse exits on an unknown flag
        assert a.no_also and a.no_gather and a.no_end_to_end and a.gpus == 1
        assert "WORLD_SIZE" not in env and "RANK" not in env
        calls.append(a)
        if a.workload == "vardct8k_pq":
            return R(3, "")
        line = {"value": 1.0, "unit": "Mpixels/s", "ms_per_step": 2.0, "dtype": "int32", "config": {"workload": a.workload},
                "roofline": {"frac": 0.25, "traffic": 123, "traffic_source": "profiles/x.json"}, "cpu_baseline": {"value": 9.0}}
        if a.workload == "vardct4k":
            line["roofline"]["path_frac"] = 0.1
        return R(0, "noise\n" + json.dumps(line) + "\n")

    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    monkeypatch.setenv("WORLD_SIZE", "1")
    monkeypatch.setenv("RANK", "0")
    out = bench.also_runs(bench.parse([]))
    assert len(calls) == 6 and set(keys) <= set(out)
    assert "error" in out["vardct8k_pq"] and "value" not in out["vardct8k_pq"]
    assert out["modular8k"]["frac"] == 0.25 and out["modular8k"]["traffic"] == 123 and out["modular8k"]["cpu_baseline"] == {"value": 9.0}
    assert out["vardct4k_epf1"]["frac"] == 0.1 and out["vardct4k_epf1"]["kernel_frac"] == 0.25

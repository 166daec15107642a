"""The multi-rank path of bench.py as far as ONE GPU allows: --force-dist initialises torch.distributed
over RCCL with a single rank and runs the gather legs. Checked: the payloads of the f32 and the RGB16 legs (8 frames x 3 planes
x 4 bytes = 796 MB against 8 x 3 x 2 = 398 MB: the round-2 run reported 796.3 for both) and that the gathered tensor holds what
jxl_vardct_read_output returns frame by frame."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_force_dist_gather_payloads():
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--force-dist", "--steps", "2", "--warmup", "1",
                        "--no-cpu-baseline", "--no-end-to-end"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    g = line["gather"]
    assert "error" not in g, g
    px = 3840 * 2160
    assert abs(g["f32"]["payload_MB_per_rank"] - 8 * 3 * 4 * px / 1e6) < 1.0, g["f32"]
    assert abs(g["rgb16"]["payload_MB_per_rank"] - 8 * 3 * 2 * px / 1e6) < 1.0, g["rgb16"]
    assert g["f32"]["gathered_equals_read_output"] is True and g["rgb16"]["gathered_equals_read_output"] is True
    assert line["timing"]["repetitions"] >= 5 and line["timing"]["timed_s_total"] > 0

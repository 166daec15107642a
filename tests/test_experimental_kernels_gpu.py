"""The launch-plan / item-order switches of the IDCT stage that DESIGN.md and profiles/r3_experiments.md name (each a path that was
the default at some point) stay under the same bit-exact parity bar as the product path: every frame below is decoded by a fresh
process with the switch set (the library reads it once) and compared with the oracle bit for bit. (The two round-3 kernels that
lost -- k_idct_wave, k_restore_stream -- left the library in round 5: profiles/experiments/r5_removed_r3_kernels.diff.)"""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

SCRIPT = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from jxlatte_amd import _lib, abi, host, synth
from oracle import pyoracle as orc
ctx = _lib.Context(0)
bad = 0
for mix, al, seed, size in (("default", True, 2, (512, 256)), ("all", False, 3, (512, 256)), ("dct8", True, 4, (264, 136)), ("default", True, 7, (1000, 520))):
    frame = synth.make_vardct_frame(size[0], size[1], seed=seed, mix=mix, aligned=al)
    for st in (1, 15):
        fr = host.Frame.from_synth(ctx, frame, stages=st)
        got = fr.decodeFrame()
        exp = orc.vardct_frame(frame, stages=st)
        n = int((got.view(np.uint32) != exp.view(np.uint32)).sum())
        print(mix, al, size, st, "launches", fr.lastLaunchCount(), "mismatch", n)
        bad += n
print("RESULT", bad)
sys.exit(1 if bad else 0)
""" % ROOT


@pytest.mark.parametrize("switch", ["JXL_WG3_LLF_IN_ITEM=0", "JXL_WG3_BALANCE=0", "JXL_WG3_BIG_FIRST=0",
                                    "JXL_WG3_BIG_AFTER", "JXL_WG3_SPATIAL=0"])
def test_switched_kernel_is_bit_exact(switch):
    """the launch-plan / item-order switches DESIGN.md and profiles/r3_experiments.md name (each a
    path that was the default at some point): same bits as the oracle"""
    env = dict(os.environ)
    name, _, val = switch.partition("=")
    env[name] = val or "1"
    r = subprocess.run([sys.executable, "-c", SCRIPT], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "RESULT 0" in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]

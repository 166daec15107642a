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


EPF3_SCRIPT = r"""
import sys
sys.path.insert(0, %r)
import numpy as np
from jxlatte_amd import _lib, abi, host, synth
from oracle import pyoracle as orc
ctx = _lib.Context(0)
bad = 0
# sizes around the tile geometry of both forms (58x26 / 64x64 / 62x30 tiles, mirrored frame edges), Gaborish on and off, float and
# integer sinks, the batch entry (which must fall back to per-frame launches for the split form)
for size, gab, seed in (((72, 40), True, 1), ((136, 72), False, 2), ((520, 264), True, 3), ((1000, 520), True, 4), ((264, 1000), False, 5)):
    frame = synth.make_vardct_frame(size[0], size[1], seed=seed, mix="default", aligned=False, epf_iters=3, gab=gab)
    for st, fmt in ((7, None), (31, abi.OUT_RGB8), (31, abi.OUT_U16)):
        p = abi.VarDCTParams.from_buffer_copy(frame["params"])
        p.stages = st
        if fmt is not None:
            p.transfer, p.out_format = abi.TRANSFER_SRGB, fmt
        f2 = dict(frame); f2["params"] = bytes(p)
        fr = host.Frame.from_synth(ctx, f2, stages=st)
        got = fr.decodeFrame()
        exp = orc.vardct_frame(f2, stages=st)
        if exp.shape != got.shape:
            exp = np.ascontiguousarray(np.moveaxis(exp, 0, -1)) if got.ndim == 3 and got.shape[-1] == 3 else exp
        n = int((np.asarray(got).view(np.uint8) != np.asarray(exp, got.dtype).view(np.uint8)).sum())
        print(size, gab, st, fmt, "launches", fr.lastLaunchCount(), "mismatch", n)
        bad += n
ctxs = [_lib.Context(0) for _ in range(3)]
frames = [synth.make_vardct_frame(264, 136, seed=20 + i, mix="default", epf_iters=3) for i in range(3)]
frs = [host.Frame.from_synth(c, f, stages=15) for c, f in zip(ctxs, frames)]
host.Frame.runBatch(frs)
for fr, f in zip(frs, frames):
    n = int((fr.readOutput().view(np.uint32) != orc.vardct_frame(f, stages=15).view(np.uint32)).sum())
    print("batch", n)
    bad += n
print("RESULT", bad)
sys.exit(1 if bad else 0)
""" % ROOT


@pytest.mark.parametrize("split", ["0", "1"])
def test_three_epf_iterations_as_one_launch_and_as_two(split):
    """JXL_EPF3_SPLIT: the 13-tap iteration in the fused launch (r4) or as a launch of its own in front of the two-iteration kernel
    (r5, default) -- same bits as the oracle either way"""
    env = dict(os.environ, JXL_EPF3_SPLIT=split)
    r = subprocess.run([sys.executable, "-c", EPF3_SCRIPT], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "RESULT 0" in r.stdout, r.stdout[-2500:] + r.stderr[-2000:]

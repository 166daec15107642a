"""GPU parity of the VarDCT path: HIP kernels (through the C-ABI) vs the CPU oracle, bit-exact.

Bar (BASELINE.json north_star): VarDCT/XYB within 1 ulp float -- we hold the stronger bar, bit
identity with the oracle, for every stage up to and including invertXYB; only the PQ/sRGB transfer
stage (double pow) is allowed 1 ulp.
"""
import numpy as np
import pytest

from conftest import assert_bits_equal
from jxlatte_amd import abi, host, synth

pytestmark = pytest.mark.gpu

STAGE_SETS = [
    ("idct", abi.STAGE_IDCT),
    ("idct+gab", abi.STAGE_IDCT | abi.STAGE_GAB),
    ("idct+gab+epf", abi.STAGE_IDCT | abi.STAGE_GAB | abi.STAGE_EPF),
    ("all-f32", abi.STAGE_IDCT | abi.STAGE_GAB | abi.STAGE_EPF | abi.STAGE_XYB),
]


def run_both(ctx, orc, frame, stages):
    fr = host.Frame.from_synth(ctx, frame, stages=stages)
    got = fr.decodeFrame()
    exp = orc.vardct_frame(frame, stages=stages)
    return got, exp


@pytest.mark.parametrize("mix,aligned,seed", [("all", True, 1), ("default", True, 2), ("all", False, 3), ("dct8", True, 4)])
@pytest.mark.parametrize("name,stages", STAGE_SETS)
def test_frame_parity_small(ctx, orc, mix, aligned, seed, name, stages):
    frame = synth.make_vardct_frame(512, 256, seed=seed, mix=mix, aligned=aligned)
    got, exp = run_both(ctx, orc, frame, stages)
    assert_bits_equal(got, exp, "%s %s" % (mix, name))


def test_frame_parity_large_blocks(ctx, orc):
    frame = synth.make_vardct_frame(1024, 512, seed=5, mix="large")
    assert any(t >= 21 for t in frame["block_types"])
    got, exp = run_both(ctx, orc, frame, abi.STAGE_IDCT)
    assert_bits_equal(got, exp, "large idct")
    got, exp = run_both(ctx, orc, frame, abi.STAGE_ALL & ~abi.STAGE_OUT)
    assert_bits_equal(got, exp, "large all")


@pytest.mark.parametrize("w,h", [(8, 8), (24, 40), (264, 520), (2056, 64)])
def test_frame_parity_ragged_sizes(ctx, orc, w, h):
    """frames that are not multiples of the 256 group / 64 tile / 2048 LF group"""
    frame = synth.make_vardct_frame(w, h, seed=w * 7 + h, mix="default")
    got, exp = run_both(ctx, orc, frame, abi.STAGE_ALL & ~abi.STAGE_OUT)
    assert_bits_equal(got, exp, "ragged %dx%d" % (w, h))


@pytest.mark.parametrize("iters", [0, 1, 2, 3])
def test_frame_parity_epf_iterations(ctx, orc, iters):
    frame = synth.make_vardct_frame(256, 128, seed=11 + iters, mix="default", epf_iters=iters)
    got, exp = run_both(ctx, orc, frame, abi.STAGE_ALL & ~abi.STAGE_OUT)
    assert_bits_equal(got, exp, "epf iters %d" % iters)

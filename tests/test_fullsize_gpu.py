"""BASELINE.json's full sizes on the GPU: exact comparison on a sample of groups + size-independent
properties (the oracle needs seconds per 4K frame, so full-frame comparison is done once at 4K)."""
import os

import numpy as np
import pytest

from conftest import assert_bits_equal
from jxlatte_amd import abi, host, synth

pytestmark = pytest.mark.gpu


def test_4k_vardct_full_frame_bit_exact(ctx, orc):
    """config C3: 3840x2160, mixed varblocks, Gab + EPF + XYB"""
    frame = synth.make_vardct_frame(3840, 2160, seed=1234, mix="default")
    got = host.Frame.from_synth(ctx, frame).decodeFrame()
    exp = orc.vardct_frame(frame, threads=os.cpu_count())
    assert_bits_equal(got, exp, "4K frame")


def test_4k_rerun_is_idempotent_and_group_order_free(ctx):
    """run() twice gives identical planes; feeding the groups in reverse order gives identical planes"""
    frame = synth.make_vardct_frame(3840, 2160, seed=1001, mix="default")
    fr = host.Frame.from_synth(ctx, frame)
    a = fr.decodeFrame()
    b = fr.decodeFrame()
    assert_bits_equal(a, b, "rerun")
    p = abi.VarDCTParams.from_buffer_copy(frame["params"])
    fr2 = host.Frame(ctx, p, frame["weights"], frame["woffs"])
    for g in reversed(frame["lfgroups"]):
        fr2.setLFGroup(g)
    for grp in reversed(range(synth.num_groups(frame))):
        fr2.putGroup(0, grp, synth.group_view(frame, grp))
    assert_bits_equal(fr2.decodeFrame(), a, "reverse order")


def test_large_block_mix_frame(ctx, orc):
    """128/256-edge varblocks across several LF groups (2048-px boundary crossed)"""
    frame = synth.make_vardct_frame(2304, 512, seed=9, mix="large")
    assert any(t >= 24 for t in frame["block_types"]) and len(frame["lfgroups"]) == 2
    got = host.Frame.from_synth(ctx, frame).decodeFrame()
    assert_bits_equal(got, orc.vardct_frame(frame, threads=os.cpu_count()), "large mix")


def test_lf_only_frame_is_flat_at_8k(ctx):
    """size-independent property at 8K: zero HF coefficients + constant LF -> constant planes after the
    IDCT stage for every varblock type, Gab keeps a constant, EPF keeps a constant"""
    frame = synth.make_vardct_frame(7680, 4320, seed=4321, mix="default", nonzero_p=0.0, xyb=False)
    for g in frame["lfgroups"]:
        for c in range(3):
            g["lf"][c][:] = np.float32(0.25) * (c + 1)
    out = host.Frame.from_synth(ctx, frame).decodeFrame()
    for c in range(3):
        assert np.abs(out[c] - np.float32(0.25) * (c + 1)).max() < 2e-6


def test_8k_modular_segmented_equals_serial_walk(ctx, monkeypatch):
    """BASELINE's 8K Modular size through a size-independent property: the segmented squeeze, with either form of the
    horizontal step or the size-dependent mix of both, gives exactly what one serial walk per row / column gives"""
    mod = synth.make_modular_frame(7680, 4320, channels=3, seed=7)
    outs = {}
    for mode, env in (("serial", {"JXL_SQUEEZE_SERIAL": "1"}), ("lds", {"JXL_HSQUEEZE_WALK_MAX": "0"}),
                      ("hybrid", {}), ("walk", {"JXL_HSQUEEZE_WALK_MAX": str(1 << 40)})):
        for k in ("JXL_SQUEEZE_SERIAL", "JXL_HSQUEEZE_WALK_MAX"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ms = host.ModularStream(ctx, mod["chans"], mod["sp"])
        outs[mode] = [np.array(a, copy=True) for a in ms.applyTransforms()]
    for mode in ("lds", "hybrid", "walk"):
        for a, b in zip(outs[mode], outs["serial"]):
            assert a.shape == (4320, 7680) and np.array_equal(a, b), mode

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


def test_c5_batch_of_4k_frames(orc):
    """config C5's per-GPU share: 8 independent 4K frames (seeds 1000..1007, the frames rank 0 of 8 holds for
    frames-per-gpu 8 are 1000, 1008, ...; the generator is the same) through per-frame runs on 8 contexts / streams and
    through jxl_vardct_run_batch: identical planes both ways, two of them checked against the oracle"""
    from jxlatte_amd import _lib
    seeds = list(range(1000, 1008))
    ctxs = [_lib.Context(0) for _ in seeds]
    try:
        synths = {}
        frames = []
        for c, sd in zip(ctxs, seeds):
            fr = synth.make_vardct_frame(3840, 2160, seed=sd, mix="default")
            if sd in (1000, 1005):
                synths[sd] = fr
            frames.append(host.Frame.from_synth(c, fr))
        for fr in frames:          # what bench.py's step does
            fr.run()
        single = [fr.readOutput() for fr in frames]
        host.Frame.runBatch(frames)
        for i, fr in enumerate(frames):
            assert_bits_equal(fr.readOutput(), single[i], "frame %d: batch vs single run" % seeds[i])
        for sd, f in synths.items():
            assert_bits_equal(single[seeds.index(sd)], orc.vardct_frame(f, threads=os.cpu_count()), "frame %d vs oracle" % sd)
        assert not np.array_equal(single[0], single[1])  # distinct frames really are distinct
    finally:
        for c in ctxs:
            c.close()


def test_c4_8k_pq_u16_vs_oracle(ctx, orc):
    """config C4 at its full size: 7680x4320, BT.2100-adapted opsin matrix, XYB -> linear -> PQ -> u16, against the oracle
    (double-precision pow, TransferFunction.java:83-87): EVERY 16-bit code value identical (r3: PQ + quantisation through the
    table and the 65 535 thresholds of the composite, fp_pq16 -- exact for all 2^32 inputs, profiles/r3_pq16_sweep.txt; until r3
    the float route was within one code value with 1 sample in ~10^4 off by one)"""
    frame = synth.make_vardct_frame(7680, 4320, seed=4321, mix="default", transfer=abi.TRANSFER_PQ, out_format=abi.OUT_U16,
                                    opsin_matrix=synth.bt2100_opsin_matrix(), intensity_target=10000.0)
    got = host.Frame.from_synth(ctx, frame).decodeFrame()
    exp = orc.vardct_frame(frame, threads=os.cpu_count())
    assert got.shape == exp.shape == (3, 4320, 7680)
    d = np.abs(got.astype(np.int32) - exp.astype(np.int32))
    assert d.max() == 0, "PQ u16: %d samples differ, by up to %d code values" % (int((d != 0).sum()), d.max())


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


def test_8k_modular_default_plan_vs_oracle(ctx, orc):
    """north_star's 8K Modular size against the ORACLE (ModularChannel.java:361-413 over the default plan of
    ModularStream.java:110-131,229-254), not only against the device's own serial walk: 7680 x 4320 x 3, seed 7, the library's own
    segment / chunk-width choice (k_inv_vh32 at a pitch that is a multiple of 128 bytes -- where kVhPad's line alignment bites and where
    the r5 store-data hazard showed first). Twice: that hazard was run-to-run."""
    mod = synth.make_modular_frame(7680, 4320, channels=3, seed=7)
    exp = orc.modular_apply(mod["chans"], mod["sp"])
    ms = host.ModularStream(ctx, mod["chans"], mod["sp"])
    for rep in range(2):
        got = ms.applyTransforms()
        assert len(got) == len(exp) == 3
        for i, (a, b) in enumerate(zip(got, exp)):
            assert a.shape == b.shape == (4320, 7680)
            assert np.array_equal(a, b), "8K modular channel %d (run %d): %d samples differ" % (i, rep, int((a != b).sum()))

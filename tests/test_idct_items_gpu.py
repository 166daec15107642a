"""r6: the item kinds of the persistent inverse-transform launch (k_idct_wg3.hip) against the oracle, case by case -- the special 8x8
types (Hornuss, DCT2, DCT4, DCT4x8, DCT8x4, AFV0-3: PassGroup.java:88-168, 234-325) and the 64x64 blocks (Item64) as items, the explicit
per-workgroup item lists with holes (wg3_item_table), and the switches that put them back into launches of their own. Bit compare through
the C-ABI, as everywhere (tests/conftest.py)."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from jxlatte_amd import _lib, abi, host, synth

pytestmark = pytest.mark.gpu

SPECIALS = ["HORNUSS", "DCT2", "DCT4", "DCT4_8", "DCT8_4", "AFV0", "AFV1", "AFV2", "AFV3"]


def _both(ctx, orc, frame, stages=abi.STAGE_IDCT):
    got = host.Frame.from_synth(ctx, frame, stages=stages).decodeFrame()
    return got, orc.vardct_frame(frame, stages=stages)


@pytest.mark.parametrize("name", SPECIALS + ["DCT64"])
@pytest.mark.parametrize("size", [(64, 64), (328, 200), (1024, 520)])
def test_frame_of_one_item_kind(ctx, orc, name, size):
    """a frame of ONE special type / of 64x64 blocks (every item of the launch is of the new kind: each special item sits behind a
    special item, each 64x64 item behind a 64x64 item), at a size smaller than one item, a ragged one and one with many items"""
    frame = synth.make_vardct_frame(size[0], size[1], seed=len(name) * 131 + size[0], mix="%s=1.0" % name)
    hist = synth.type_histogram(frame)
    assert hist.get(name, 0) > 0.5 or size == (328, 200) or name == "DCT64", hist
    got, exp = _both(ctx, orc, frame)
    assert_bits_equal(got, exp, "%s %s" % (name, size))


@pytest.mark.parametrize("aligned", [True, False])
def test_mix_with_every_item_kind_and_unaligned_blocks(ctx, orc, aligned):
    """all 27 types in one frame, aligned and unaligned tilings (a 64x64 block that straddles 64x64 chroma-from-luma tiles takes its
    factors per group of four samples, with the reference's cache-order mask), whole path"""
    frame = synth.make_vardct_frame(1280, 768, seed=77, mix="all", aligned=aligned)
    assert any(t == 18 for t in frame["block_types"]) and any(t in (1, 2, 3, 12, 13, 14, 15, 16, 17) for t in frame["block_types"])
    for stages in (abi.STAGE_IDCT, abi.STAGE_ALL & ~abi.STAGE_OUT):
        got, exp = _both(ctx, orc, frame, stages)
        assert_bits_equal(got, exp, "all types, aligned=%s, stages %d" % (aligned, stages))


@pytest.mark.parametrize("mix", ["DCT64=0.5+AFV1=0.25+DCT8=0.25", "DCT64=1.0", "HORNUSS=0.5+DCT32=0.5"])
def test_large_multipliers_and_coefficients(ctx, orc, mix):
    """hfMultiplier values beyond the per-workgroup quotient table (>= 256: the in-place division) and |q| >= 64 (beyond the
    dequantisation table: the wave-uniform slow branch), for the new item kinds too"""
    frame = synth.make_vardct_frame(512, 384, seed=5, mix=mix, coeff_scale=300.0, nonzero_p=0.4)
    assert max(int(np.abs(c).max()) for c in frame["coeff"]) >= 64
    rng = np.random.default_rng(9)
    for g in frame["lfgroups"]:
        m = np.array(g["hf_mul"], copy=True)
        m[...] = rng.choice(np.array([1, 7, 255, 256, 300, 4097], np.int32), size=m.shape)
        # one multiplier per varblock (HFMetadata: hfMultiplier is per block): copy the block origin's value over its cells
        g["hf_mul"] = m
    # make the multiplier constant inside every varblock (the reference stores it per block)
    _uniform_per_block(frame)
    got, exp = _both(ctx, orc, frame)
    assert_bits_equal(got, exp, "large multipliers, %s" % mix)


def _uniform_per_block(frame):
    for g in frame["lfgroups"]:
        m = g["hf_mul"]
        for (by, bx), t in zip(np.asarray(g["block_yx"]).reshape(-1, 2), _types_of(g)):
            ph, pw = abi.tt_pixel_size(int(t))
            m[by:by + ph // 8, bx:bx + pw // 8] = m[by, bx]


def _types_of(g):
    sel = np.asarray(g["dct_select"])
    return [sel[by, bx] for by, bx in np.asarray(g["block_yx"]).reshape(-1, 2)]


@pytest.mark.parametrize("env", [{"JXL_WG3_SPECIAL": "0"}, {"JXL_WG3_FOLD64": "0"}, {"JXL_WG3_SPECIAL": "0", "JXL_WG3_FOLD64": "0"},
                                 {"JXL_WG3_GRID": "8"}, {"JXL_WG3_GRID": "40"}, {"JXL_WG3_GRID": "2048"}, {"JXL_WG3_BALANCE": "0"},
                                 {"JXL_WG3_SPATIAL": "0"}, {"JXL_WG3_LLF_IN_ITEM": "0"}, {"JXL_WG3_LLF_IN_ITEM": "0", "JXL_WG3_FOLD64": "0"}])
def test_switches_give_identical_planes(env, orc):
    """the r5 launch plan (special kernel, 64-point class's own launch), other persistent grids (8: a workgroup walks 1/8 of the frame;
    40: five workgroups per queue; 2048: more workgroups than can be resident), no balancing, no spatial order, finalizeLLF as a launch of
    its own in front (the items then take their LLF corner from the llf planes): the same bits. The switches are read once per process, hence a process per case."""
    import os
    import subprocess
    import sys
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
from jxlatte_amd import _lib, abi, host, synth
from oracle import pyoracle as orc
ctx = _lib.Context(0)
ok = True
for seed, size, mix in ((3, (1024, 640), "all"), (4, (520, 264), "default"), (5, (256, 256), "DCT64=0.6+AFV2=0.4")):
    fr = synth.make_vardct_frame(size[0], size[1], seed=seed, mix=mix, aligned=seed != 4)
    got = host.Frame.from_synth(ctx, fr, stages=abi.STAGE_IDCT).decodeFrame()
    exp = orc.vardct_frame(fr, stages=abi.STAGE_IDCT)
    ok = ok and np.array_equal(got.view(np.uint32), exp.view(np.uint32))
print("LAUNCHES", ctx.lib.jxl_vardct_last_launch_count(ctx.h))
print("OK" if ok else "MISMATCH")
""" % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([sys.executable, "-c", code], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "OK" in r.stdout.split(), (env, r.stdout[-500:], r.stderr[-1500:])


def test_default_mix_is_one_idct_launch(ctx):
    """what the round was about: a frame of the default mix -- METHOD_DCT up to 64x64 and the nine special types -- is ONE inverse-transform
    launch (r5: three); 64x32 blocks add the 512-thread class's launch"""
    frame = synth.make_vardct_frame(1024, 512, seed=1, mix="default")
    fr = host.Frame.from_synth(ctx, frame, stages=abi.STAGE_IDCT)
    fr.run()
    assert fr.lastLaunchCount() == 1
    frame = synth.make_vardct_frame(1024, 512, seed=1, mix="DCT8=0.5+DCT64_32=0.5")
    fr = host.Frame.from_synth(ctx, frame, stages=abi.STAGE_IDCT)
    fr.run()
    assert fr.lastLaunchCount() == 2


def test_run_batch_walks_the_lists_on_another_grid(orc):
    """jxl_vardct_run_batch launches the frames' item tables on a grid of its own choosing: the walk steps over the holes, every item is
    done exactly once"""
    seeds = [21, 22, 23]
    ctxs = [_lib.Context(0) for _ in seeds]
    try:
        frames, synths = [], []
        for c, sd in zip(ctxs, seeds):
            f = synth.make_vardct_frame(768, 512, seed=sd, mix="DCT64=0.3+AFV0=0.2+DCT16=0.2+DCT8=0.3")
            synths.append(f)
            frames.append(host.Frame.from_synth(c, f))
        host.Frame.runBatch(frames)
        for fr, f in zip(frames, synths):
            assert_bits_equal(fr.readOutput(), orc.vardct_frame(f), "batch frame")
    finally:
        for c in ctxs:
            c.close()

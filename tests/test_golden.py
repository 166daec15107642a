"""Golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from the CPU oracle).

CPU part: today's oracle still reproduces the committed vectors (so a change in the oracle cannot
silently move the goal posts). GPU part: the HIP path reproduces them bit for bit."""
import os

import numpy as np
import pytest

from conftest import assert_bits_equal
from jxlatte_amd import abi, host

HERE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
import sys
sys.path.insert(0, HERE)
from make_golden import npz_to_frame  # noqa: E402

VARDCT = ["vardct_aligned", "vardct_unaligned", "vardct_epf3_nogab"]
STAGES = {"idct": 1, "gab": 3, "epf": 7, "xyb": 15}


def load(name):
    return np.load(os.path.join(HERE, name + ".npz"))


@pytest.mark.parametrize("name", VARDCT)
def test_oracle_reproduces_vardct_goldens(orc, name):
    z = load(name)
    fr = npz_to_frame(z)
    for tag, st in STAGES.items():
        if "expect_" + tag in z:
            assert_bits_equal(orc.vardct_frame(fr, stages=st), z["expect_" + tag], "%s %s" % (name, tag))


def test_oracle_reproduces_stage_goldens(orc):
    z = load("stages")
    p = z["planes"]
    assert_bits_equal(orc.gab(p, [0.115169525] * 3, [0.061248592] * 3), z["gab"], "gab")
    for it in (1, 2, 3):
        assert_bits_equal(orc.epf(p, it, z["inv_sigma"], 0.0, (40.0, 5.0, 3.5), 0.9, 6.5, 2.0 / 3.0), z["epf%d" % it], "epf%d" % it)
    assert_bits_equal(orc.idct2d(z["dct_in"]), z["idct_32x64"], "idct")
    assert_bits_equal(orc.idct2d(z["dct_in"], transposed=True), z["idct_32x64_t"], "idct_t")
    assert_bits_equal(orc.fdct2d(z["dct_in"]), z["fdct_32x64"], "fdct")


def test_oracle_reproduces_modular_golden(orc):
    z = load("modular_53x37")
    sp = [tuple(int(v) for v in r) for r in z["sp"]]
    chans = [z["chan%d" % i] for i in range(len([k for k in z.files if k.startswith("chan")]))]
    out = orc.modular_apply(chans, sp, rct_type=int(z["rct_type"]), rct_begin=0)
    for i, o in enumerate(out):
        assert_bits_equal(o, z["out%d" % i], "modular out%d" % i)


def test_oracle_reproduces_post_golden(orc):
    from make_golden import post_vectors
    z = load("post")
    again = post_vectors({k: z[k] for k in ("plane", "up_packed2", "up_packed4", "xyb", "noise_lut", "frame", "frame_alpha", "ref",
                                            "ref_alpha", "ints")})
    assert set(again) == set(z.files)
    for k in z.files:
        assert_bits_equal(again[k], z[k], "post " + k)


# ---- GPU -----------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("name", VARDCT)
def test_hip_reproduces_vardct_goldens(ctx, name):
    z = load(name)
    fr = npz_to_frame(z)
    for tag, st in STAGES.items():
        if "expect_" + tag in z:
            got = host.Frame.from_synth(ctx, fr, stages=st).decodeFrame()
            assert_bits_equal(got, z["expect_" + tag], "%s %s" % (name, tag))
    if "expect_srgb_u8" in z:
        fr["params"].transfer, fr["params"].out_format = abi.TRANSFER_SRGB, abi.OUT_U8
        got = host.Frame.from_synth(ctx, fr, stages=31).decodeFrame()
        assert got.dtype == np.uint8
        # sRGB uses a double pow (1 ulp allowed by the parity bar): quantised values may differ by 1 LSB at most
        d = np.abs(got.astype(np.int32) - z["expect_srgb_u8"].astype(np.int32))
        assert d.max() <= 1 and (d != 0).mean() < 1e-3


@pytest.mark.gpu
def test_hip_reproduces_stage_goldens(ctx):
    z = load("stages")
    p = z["planes"]
    assert_bits_equal(host.performGabConvolution(ctx, p, [0.115169525] * 3, [0.061248592] * 3), z["gab"], "gab")
    for it in (1, 2, 3):
        got = host.performEdgePreservingFilter(ctx, p, it, z["inv_sigma"])
        assert_bits_equal(got, z["epf%d" % it], "epf%d" % it)
    assert_bits_equal(host.MathHelper.inverseDCT2D(ctx, z["dct_in"]), z["idct_32x64"], "idct")
    assert_bits_equal(host.MathHelper.inverseDCT2D(ctx, z["dct_in"], True), z["idct_32x64_t"], "idct_t")
    assert_bits_equal(host.MathHelper.forwardDCT2D(ctx, z["dct_in"]), z["fdct_32x64"], "fdct")


@pytest.mark.gpu
def test_hip_reproduces_modular_golden(ctx):
    z = load("modular_53x37")
    sp = [tuple(int(v) for v in r) for r in z["sp"]]
    n = len([k for k in z.files if k.startswith("chan")])
    ms = host.ModularStream(ctx, [z["chan%d" % i] for i in range(n)], sp, rctType=int(z["rct_type"]), rctBegin=0)
    out = ms.applyTransforms()
    assert len(out) == 3
    for i, o in enumerate(out):
        assert_bits_equal(o, z["out%d" % i], "modular out%d" % i)


@pytest.mark.gpu
def test_hip_reproduces_post_golden(ctx):
    """rows f4 / f3 against the committed fixture (no oracle involved)"""
    from make_golden import POST_RECT, POST_SEED
    z = load("post")
    pl = z["plane"]
    assert_bits_equal(host.invertSubsampling(ctx, pl, 1, 1), z["chroma_11"], "chroma 1,1")
    assert_bits_equal(host.invertSubsampling(ctx, pl, 2, 0), z["chroma_20"], "chroma 2,0")
    for k in (2, 4):
        assert_bits_equal(host.performUpsampling(ctx, pl, k, host.getUpWeights(k, z["up_packed%d" % k])), z["up%d" % k], "up%d" % k)
    nz = host.initializeNoise(ctx, 20, 28, POST_SEED, groupDim=16)
    assert_bits_equal(nz, z["noise"], "noise")
    assert_bits_equal(host.synthesizeNoise(ctx, z["xyb"], nz, z["noise_lut"], 0.0, 1.0), z["noise_added"], "noise added")
    for mode, kw in ((abi.BLEND_ADD, {}), (abi.BLEND_MULT, dict(clamp=True)), (abi.BLEND_BLEND, dict(hasExtra=True, clamp=True)),
                     (abi.BLEND_BLEND, dict(hasExtra=True, premult=True)), (abi.BLEND_MULADD, dict(hasExtra=True))):
        got = host.blend(ctx, mode, z["ref"], z["frame"], z["ref"], POST_RECT, frameAlpha=z["frame_alpha"], refAlpha=z["ref_alpha"], **kw)
        assert_bits_equal(got, z["blend_%d_%d" % (mode, 1 if kw.get("premult") else 0)], "blend %d" % mode)
    assert_bits_equal(host.transposeBuffer(ctx, z["ints"][0], 6), z["orient6"], "orient 6")
    assert_bits_equal(host.transposeBuffer(ctx, pl, 7), z["orient7"], "orient 7")
    assert_bits_equal(host.packSamples(ctx, list(z["xyb"]), 8), z["pack_rgb8"], "pack rgb8")
    got = host.packSamples(ctx, list(z["ints"]), 16, alpha=z["ref_alpha"], premultiplied=True, taggedDepth=[12, 12, 12, 8], bigEndian=True)
    assert_bits_equal(got, z["pack_rgba16be"], "pack rgba16be")

"""GPU parity of every stage-level C-ABI entry point against the oracle, bit-exact (1 ulp for the
double-pow transfer functions), including the edge cases the reference's loops have: 1-pixel planes,
sizes that are not multiples of 8, mirrored/clamped borders, inf/NaN sigma."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from jxlatte_amd import _lib, abi, host, synth

pytestmark = pytest.mark.gpu
F = np.float32


@pytest.mark.parametrize("h,w,t", [(1, 1, False), (1, 8, False), (8, 1, True), (4, 4, True), (4, 8, False), (8, 8, False), (16, 32, True),
                                   (64, 64, False), (128, 64, False), (256, 256, False), (256, 128, True)])
def test_idct2d_fdct2d(ctx, orc, h, w, t):
    x = np.random.default_rng(h * 7 + w).standard_normal((h, w)).astype(F)
    assert_bits_equal(host.MathHelper.inverseDCT2D(ctx, x, t), orc.idct2d(x, t), "idct %dx%d" % (h, w))
    assert_bits_equal(host.MathHelper.forwardDCT2D(ctx, x), orc.fdct2d(x), "fdct %dx%d" % (h, w))


def test_library_lut_equals_oracle_lut(ctx, orc):
    """IDCT of unit impulses exposes the device LUT rows: must equal the oracle's table for every size"""
    for l in range(1, 9):
        n = 1 << l
        lut = orc.cosine_lut(l)
        for j in (1, n // 2, n - 1):
            x = np.zeros((1, n), F)
            x[0, j] = 1.0
            got = host.MathHelper.inverseDCT2D(ctx, x)
            assert_bits_equal(got[0], lut[j - 1], "lut size %d row %d" % (n, j))


def test_idct2d_rejects_bad_size(ctx):
    with pytest.raises(_lib.IllegalArgumentException):
        host.MathHelper.inverseDCT2D(ctx, np.zeros((3, 8), F))


@pytest.mark.parametrize("h,w", [(1, 1), (1, 9), (9, 1), (8, 8), (37, 53), (64, 130), (270, 480)])
def test_gab(ctx, orc, h, w):
    p = np.random.default_rng(h + w).standard_normal((3, h, w)).astype(F)
    w1, w2 = [0.115169525, 0.2, 0.05], [0.061248592, 0.01, 0.1]
    assert_bits_equal(host.performGabConvolution(ctx, p, w1, w2), orc.gab(p, w1, w2), "gab %dx%d" % (h, w))


@pytest.mark.parametrize("h,w", [(1, 1), (2, 3), (8, 8), (13, 29), (64, 72), (135, 240)])
@pytest.mark.parametrize("iters", [0, 1, 2, 3])
def test_epf_with_sigma_map(ctx, orc, h, w, iters):
    rng = np.random.default_rng(h * 31 + w + iters)
    p = (rng.standard_normal((3, h, w)) * 0.1).astype(F)
    sig = (rng.random(((h + 7) // 8, (w + 7) // 8)) * 5).astype(F)
    sig.flat[0] = np.inf
    if sig.size > 2:
        sig.flat[1] = np.nan
        sig.flat[2] = 3.4  # > 1/0.3: copied
    args = ((40.0, 5.0, 3.5), 0.9, 6.5, 2.0 / 3.0)
    got = host.performEdgePreservingFilter(ctx, p, iters, sig, 0.0, *args)
    assert_bits_equal(got, orc.epf(p, iters, sig, 0.0, *args), "epf %dx%d it%d" % (h, w, iters))


def test_epf_modular_constant_sigma(ctx, orc):
    p = (np.random.default_rng(4).standard_normal((3, 33, 47)) * 0.05).astype(F)
    got = host.performEdgePreservingFilter(ctx, p, 2, None, 1.0, (20.0, 4.0, 1.5), 0.8, 5.0, 0.5)
    assert_bits_equal(got, orc.epf(p, 2, None, 1.0, (20.0, 4.0, 1.5), 0.8, 5.0, 0.5), "epf modular")


def test_epf_sigma_map_and_error(ctx, orc):
    rng = np.random.default_rng(8)
    hf = rng.integers(1, 20, size=(9, 13)).astype(np.int32)
    sh = rng.integers(0, 8, size=(9, 13)).astype(np.int32)
    lut = [float(v) for v in synth.default_params(8, 8).epf_sharp_lut]
    assert_bits_equal(host.epfInverseSigma(ctx, hf, sh, 26.2144, lut), orc.epf_sigma(hf, sh, 26.2144, lut), "sigma")
    sh[4, 4] = 9
    with pytest.raises(_lib.InvalidBitstreamException):  # Frame.java:565-566
        host.epfInverseSigma(ctx, hf, sh, 26.2144, lut)


def test_xyb_and_ycbcr(ctx, orc):
    p = synth.default_params(8, 8)
    m = host.OpsinInverseMatrix(list(p.opsin_matrix), list(p.opsin_bias), list(p.cbrt_opsin_bias))
    x = (np.random.default_rng(2).standard_normal((3, 37, 91)) * 0.1).astype(F)
    for it in (255.0, 10000.0, 80.0):
        assert_bits_equal(m.invertXYB(ctx, x, it), orc.xyb(x, m.matrix, m.opsinBias, m.cbrtOpsinBias, it), "xyb it=%g" % it)
    assert_bits_equal(host.performColorTransformsYCbCr(ctx, x), orc.ycbcr(x), "ycbcr")
    with pytest.raises(ValueError):
        m.invertXYB(ctx, x[:2], 255.0)


def ulp_diff(a, b):
    a = a.view(np.int32).astype(np.int64)
    b = b.view(np.int32).astype(np.int64)
    a = np.where(a < 0, -(a & 0x7fffffff), a)
    b = np.where(b < 0, -(b & 0x7fffffff), b)
    return np.abs(a - b)


@pytest.mark.parametrize("tf", [abi.TRANSFER_PQ, abi.TRANSFER_SRGB])
def test_transfer_within_one_ulp(ctx, orc, tf):
    """tolerance stated by north_star: <= 1 ulp float for the transfer stage (double pow, 1-ulp libm)"""
    rng = np.random.default_rng(tf)
    x = np.concatenate([rng.random(20000), rng.random(2000) * 1e-3, [0.0, 1.0, 0.0031306, 0.0031307, 0.5]]).astype(F)
    got, exp = host.transfer(ctx, x, tf), orc.transfer(x, tf)
    d = ulp_diff(got, exp)
    assert d.max() <= 1, d.max()
    # sRGB keeps the double-precision form: the float casts agree except for double results within 1e-13 of a rounding
    # boundary. PQ inside [2^-40, 4) is the tabulated form, within 0.51 ulp of the true value: it differs from the
    # reference's correctly rounded cast whenever the true value lies within ~0.01 ulp of a boundary, i.e. for a few per cent
    # of the inputs, never by more than 1 ulp (all 2^32 inputs: tools/pq_sweep.py, profiles/r2_pq_sweep.txt)
    assert (d != 0).mean() < (0.05 if tf == abi.TRANSFER_PQ else 1e-3)
    for maxv in (255, 65535):
        gq, eq = host.transfer(ctx, x, tf, maxv), orc.transfer(x, tf, maxv)
        assert np.abs(gq - eq).max() <= 1 and (gq != eq).mean() < 1e-3


def test_transfer_pq_exact_form(ctx, orc):
    """JXL_TRANSFER_PQ_EXACT: the double-precision PQ on the device against the oracle's libm form -- the float
    results may only differ where the double result sits within an ulp of double of a float rounding boundary"""
    rng = np.random.default_rng(99)
    x = np.concatenate([rng.random(200000), rng.random(20000) * 1e-3, rng.random(20000) * 4.0, [0.0, 1.0, 0.5]]).astype(F)
    got, exp = host.transfer(ctx, x, abi.TRANSFER_PQ_EXACT), orc.transfer(x, abi.TRANSFER_PQ)
    d = ulp_diff(got, exp)
    assert d.max() <= 1 and (d != 0).mean() < 1e-4, (d.max(), (d != 0).mean())
    gq, eq = host.transfer(ctx, x, abi.TRANSFER_PQ_EXACT, 65535), orc.transfer(x, abi.TRANSFER_PQ, 65535)
    assert np.abs(gq - eq).max() <= 1 and (gq != eq).mean() < 1e-5


@pytest.mark.parametrize("tf", [abi.TRANSFER_PQ, abi.TRANSFER_SRGB])
def test_transfer_special_values_and_range(ctx, orc, tf):
    """Math.pow semantics of the inputs a frame can produce: negative (out-of-gamut) samples give NaN through PQ, zeros,
    infinities, NaN, denormals, the largest floats; and a log-uniform sweep over the whole positive float range"""
    sp = np.array([-1e30, -1.0, -1e-20, -0.0, 0.0, 1e-45, 1e-38, 1e-30, 1e-10, 0.999999, 1.0, 1.0000001, 12.5, 1e10, 3e38,
                   np.inf, -np.inf, np.nan], F)
    got, exp = host.transfer(ctx, sp, tf), orc.transfer(sp, tf)
    assert np.array_equal(np.isnan(got), np.isnan(exp)), (got, exp)
    fin = ~np.isnan(exp)
    assert np.array_equal(np.isinf(got[fin]), np.isinf(exp[fin])) and np.array_equal(np.signbit(got[fin]), np.signbit(exp[fin]))
    ok = fin & np.isfinite(exp)
    assert ulp_diff(got[ok], exp[ok]).max() <= 1
    rng = np.random.default_rng(11 + tf)
    x = (10.0 ** rng.uniform(-44, 38.5, 200000)).astype(F)
    got, exp = host.transfer(ctx, x, tf), orc.transfer(x, tf)
    d = ulp_diff(got, exp)
    # (PQ: the 15 % of this sweep that falls into the tabulated range differs by 1 ulp in a few per cent of the cases)
    assert d.max() <= 1 and (d != 0).mean() < (1e-2 if tf == abi.TRANSFER_PQ else 1e-4), (int(d.max()), float((d != 0).mean()))


def test_pq_16bit_output_is_exact(ctx, orc):
    """PQ + 16-bit quantisation: table value + one look at the composite's thresholds (fp_pq16): the oracle's code value for every input"""
    rng = np.random.default_rng(22)
    x = np.concatenate([rng.random(1000000), 10.0 ** rng.uniform(-12, 0.5, 500000), -(10.0 ** rng.uniform(-10, 0, 50000)),
                        [0.0, -0.0, 1.0, 0.99999994, 1.0000001, 3.9, 4.0, 1e30, 3e38, np.inf, -np.inf, np.nan, 1e-45, 9e-13]]).astype(F)
    x = np.concatenate([x, rng.integers(0, 2 ** 32, 500000, dtype=np.uint64).astype(np.uint32).view(F)])
    assert np.array_equal(host.transfer(ctx, x, abi.TRANSFER_PQ, 65535), orc.transfer(x, abi.TRANSFER_PQ, 65535))


def test_pq_8bit_output_is_exact(ctx, orc):
    """PQ + 8-bit quantisation: binary search in the composite's 255 thresholds (fp_pq8)"""
    rng = np.random.default_rng(24)
    x = np.concatenate([rng.random(500000), 10.0 ** rng.uniform(-12, 0.5, 300000), -(10.0 ** rng.uniform(-10, 0, 50000)),
                        [0.0, -0.0, 1.0, 0.99999994, 1.0000001, 3.9, 4.0, 1e30, 3e38, np.inf, -np.inf, np.nan, 1e-45]]).astype(F)
    x = np.concatenate([x, rng.integers(0, 2 ** 32, 300000, dtype=np.uint64).astype(np.uint32).view(F)])
    assert np.array_equal(host.transfer(ctx, x, abi.TRANSFER_PQ, 255), orc.transfer(x, abi.TRANSFER_PQ, 255))


def test_srgb_16bit_output_is_exact(ctx, orc):
    """sRGB + 16-bit quantisation: segment table + thresholds (fp_srgb16): the oracle's code value for every input"""
    rng = np.random.default_rng(23)
    x = np.concatenate([rng.random(1000000), 10.0 ** rng.uniform(-12, 0.5, 500000), -(10.0 ** rng.uniform(-10, 0, 50000)),
                        [0.0, -0.0, 1.0, 0.99999994, 1.0000001, 0.0031306684, 0.0031306685, 2.0 ** -9, 1e30, 3e38, np.inf, -np.inf, np.nan, 1e-45]]).astype(F)
    x = np.concatenate([x, rng.integers(0, 2 ** 32, 500000, dtype=np.uint64).astype(np.uint32).view(F)])
    assert np.array_equal(host.transfer(ctx, x, abi.TRANSFER_SRGB, 65535), orc.transfer(x, abi.TRANSFER_SRGB, 65535))


def test_srgb_8bit_output_is_exact(ctx, orc):
    """sRGB + 8-bit quantisation runs through the threshold table (fp_srgb8): the oracle's integer for every input, not a tolerance"""
    rng = np.random.default_rng(21)
    x = np.concatenate([rng.random(1000000), 10.0 ** rng.uniform(-10, 0.5, 500000), -(10.0 ** rng.uniform(-10, 0, 100000)),
                        [0.0, -0.0, 1.0, 0.99999994, 1.0000001, 0.0031306684, 0.0031306685, 2.0 ** -9, 1e30, np.inf, -np.inf, np.nan]]).astype(F)
    x = np.concatenate([x, rng.integers(0, 2 ** 32, 500000, dtype=np.uint64).astype(np.uint32).view(F)])
    assert np.array_equal(host.transfer(ctx, x, abi.TRANSFER_SRGB, 255), orc.transfer(x, abi.TRANSFER_SRGB, 255))


def test_quantise_java_int_cast_semantics(ctx, orc):
    x = np.array([-1e30, -0.2, -0.0, 0.0, 0.49, 0.5, 1.0, 7.0, 1e30, np.inf, -np.inf, np.nan], F)
    for maxv in (255, 65535):
        assert np.array_equal(host.transfer(ctx, x, abi.TRANSFER_NONE, maxv), orc.transfer(x, abi.TRANSFER_NONE, maxv))


def test_fused_restore_equals_stage_kernels(ctx, orc):
    """the fused tile kernel and the stage-per-kernel path are two implementations of the same reference code"""
    frame = synth.make_vardct_frame(200, 136, seed=21, mix="default")
    fused = host.Frame.from_synth(ctx, frame, stages=15).decodeFrame()
    idct = host.Frame.from_synth(ctx, frame, stages=abi.STAGE_IDCT).decodeFrame()
    p = frame["params"]
    g = host.performGabConvolution(ctx, idct, list(p.gab_w1), list(p.gab_w2))
    sig = host.epfInverseSigma(ctx, frame["hf_mul"], frame["sharpness"], p.global_scale_f, list(p.epf_sharp_lut))
    e = host.performEdgePreservingFilter(ctx, g, p.epf_iters, sig, 0.0, list(p.epf_channel_scale), p.epf_pass0_sigma_scale,
                                         p.epf_pass2_sigma_scale, p.epf_border_sad_mul)
    m = host.OpsinInverseMatrix(list(p.opsin_matrix), list(p.opsin_bias), list(p.cbrt_opsin_bias))
    assert_bits_equal(fused, m.invertXYB(ctx, e, p.intensity_target), "fused vs staged")


def test_vardct_errors(ctx):
    frame = synth.make_vardct_frame(64, 64, seed=1, mix="dct8")
    p = abi.VarDCTParams.from_buffer_copy(frame["params"])
    p.width = 63
    with pytest.raises(_lib.IllegalArgumentException):
        host.Frame(ctx, p, frame["weights"], frame["woffs"])
    bad = dict(frame)
    bad["lfgroups"] = [dict(frame["lfgroups"][0])]
    sel = bad["lfgroups"][0]["dct_select"].copy()
    sel[0, 0] = 27
    bad["lfgroups"][0]["dct_select"] = sel
    with pytest.raises(_lib.InvalidBitstreamException):  # HFMetadata.java:46-47 "Invalid Transform Type"
        host.Frame.from_synth(ctx, bad)
    bad2 = dict(frame)
    bad2["lfgroups"] = [dict(frame["lfgroups"][0])]
    sh = bad2["lfgroups"][0]["sharpness"].copy()
    sh[3, 3] = 8
    bad2["lfgroups"][0]["sharpness"] = sh
    fr = host.Frame.from_synth(ctx, bad2)
    with pytest.raises(_lib.InvalidBitstreamException):  # Frame.java:565-566
        fr.run()
    # a frame whose LF group was never provided
    fr = host.Frame(ctx, abi.VarDCTParams.from_buffer_copy(frame["params"]), frame["weights"], frame["woffs"])
    with pytest.raises(_lib.IllegalStateException):
        fr.run()


def test_progressive_passes_accumulate(ctx, orc):
    """two passes whose quantised coefficients sum to the single-pass frame give the same pixels
    (PassGroup.java:174-200; the final image equals one IDCT of the summed integers)"""
    frame = synth.make_vardct_frame(320, 264, seed=31, mix="default")
    rng = np.random.default_rng(1)
    part = (frame["coeff"] * (rng.random(frame["coeff"].shape) < 0.5)).astype(np.int32)
    rest = (frame["coeff"] - part).astype(np.int32)
    p = abi.VarDCTParams.from_buffer_copy(frame["params"])
    fr = host.Frame(ctx, p, frame["weights"], frame["woffs"])
    for g in frame["lfgroups"]:
        fr.setLFGroup(g)
    f0, f1 = dict(frame, coeff=part), dict(frame, coeff=rest)
    for grp in range(synth.num_groups(frame)):
        fr.putGroup(0, grp, synth.group_view(f0, grp))
    for grp in range(synth.num_groups(frame)):
        fr.putGroup(1, grp, synth.group_view(f1, grp))
    assert_bits_equal(fr.decodeFrame(), orc.vardct_frame(frame), "two passes")


def test_u16_pq_output_path(ctx, orc):
    frame = synth.make_vardct_frame(136, 72, seed=41, mix="default", transfer=abi.TRANSFER_PQ, out_format=abi.OUT_U16,
                                    opsin_matrix=synth.bt2100_opsin_matrix(), intensity_target=10000.0)
    got = host.Frame.from_synth(ctx, frame).decodeFrame()
    exp = orc.vardct_frame(frame)
    assert got.dtype == np.uint16
    assert np.array_equal(got, exp)  # (r3: exact -- fp_pq16)


@pytest.mark.parametrize("iters", [1, 2, 3])
def test_one_colour_gab_and_epf(ctx, orc, iters):
    """frames with one colour channel (grey Modular with restoration filters, Frame.java:638-669 `i = colors == 1 ? 0 : c`):
    the device path feeds the three-channel kernels three copies of channel 0; against the oracle's one-colour form"""
    rng = np.random.default_rng(60 + iters)
    p = (rng.standard_normal((1, 45, 83)) * 0.08).astype(F)
    g = host.performGabConvolution(ctx, p, [0.115169525, 0.3, 0.4], [0.061248592, 0.2, 0.1])
    assert g.shape == p.shape
    assert_bits_equal(g[0], orc.gab1(p[0], 0.115169525, 0.061248592), "gab one colour")
    args = ((40.0, 5.0, 3.5), 0.9, 6.5, 2.0 / 3.0)
    e = host.performEdgePreservingFilter(ctx, g, iters, None, 0.7, *args)
    assert e.shape == p.shape
    assert_bits_equal(e[0], orc.epf1(g[0], iters, None, 0.7, *args), "epf one colour it%d" % iters)

"""Row f1 of the scope table: LF dequant + LF chroma-from-luma + adaptive LF smoothing
(J/frame/vardct/LFCoefficients.java:65-180). CPU: oracle KATs; GPU: HIP vs oracle, bit-exact."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from jxlatte_amd import abi, host, synth

F = np.float32
SD = [1.0 / 4096 * 65536 / (2500 * 16), 1.0 / 512 * 65536 / (2500 * 16), 1.0 / 256 * 65536 / (2500 * 16)]  # LFGlobal.java:71


# ---- CPU: oracle ---------------------------------------------------------------------------------------
def test_oracle_dequant_and_cfl_without_smoothing(orc):
    rng = np.random.default_rng(1)
    q = rng.integers(-500, 500, size=(3, 7, 9)).astype(np.int32)
    out = orc.lf_dequant(q, SD, extra_precision=2, x_factor_lf=140, b_factor_lf=100, adaptive_smoothing=False,
                         base_corr_x=0.25, base_corr_b=1.0, color_factor=84)
    sd = [F(F(s) / F(4)) for s in SD]
    y = q[1].astype(F) * sd[1]
    kx = F(0.25) + F(F(140) - F(128)) / F(84)
    kb = F(1.0) + F(F(100) - F(128)) / F(84)
    assert_bits_equal(out[1], y, "Y")
    assert_bits_equal(out[0], (q[0].astype(F) * sd[0] + kx * y).astype(F), "X")
    assert_bits_equal(out[2], (q[2].astype(F) * sd[2] + kb * y).astype(F), "B")


def test_oracle_smoothing_keeps_constant_and_borders(orc):
    q = np.full((3, 12, 16), 321, np.int32)
    a = orc.lf_dequant(q, SD, adaptive_smoothing=True)
    b = orc.lf_dequant(q, SD, adaptive_smoothing=False)
    assert np.abs(a - b).max() < 1e-6  # kernel weights sum to 1
    rng = np.random.default_rng(2)
    q = rng.integers(-300, 300, size=(3, 12, 16)).astype(np.int32)
    a = orc.lf_dequant(q, SD, adaptive_smoothing=True)
    b = orc.lf_dequant(q, SD, adaptive_smoothing=False)
    for sl in (np.s_[:, 0, :], np.s_[:, -1, :], np.s_[:, :, 0], np.s_[:, :, -1]):
        assert_bits_equal(a[sl], b[sl], "border cells are copied")
    assert not np.array_equal(a[:, 1:-1, 1:-1], b[:, 1:-1, 1:-1])


def test_oracle_smoothing_gap_switches_off_on_strong_edges(orc):
    """|sample - smoothed| * scaledDequant >= 0.75 -> factor max(0, 3 - 4 g) = 0 -> fully smoothed value"""
    q = np.zeros((3, 9, 9), np.int32)
    q[:, 4, 4] = 4000
    big = [s * 100 for s in SD]
    a = orc.lf_dequant(q, big, adaptive_smoothing=True, x_factor_lf=128, b_factor_lf=128, base_corr_b=0.0)
    sample = F(4000) * F(big[1])
    assert a[1, 4, 4] == F(0.05226273532324128) * sample  # centre weight only: neighbours are 0


@pytest.mark.parametrize("shape", [(1, 1), (2, 2), (1, 9), (9, 2), (3, 3)])
def test_oracle_degenerate_sizes_are_copies_or_single_interior(orc, shape):
    rng = np.random.default_rng(shape[0] * 10 + shape[1])
    q = rng.integers(-300, 300, size=(3,) + shape).astype(np.int32)
    a = orc.lf_dequant(q, SD, adaptive_smoothing=True)
    b = orc.lf_dequant(q, SD, adaptive_smoothing=False)
    if min(shape) < 3:
        assert_bits_equal(a, b, "no interior")


# ---- GPU ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1), (2, 7), (3, 3), (9, 11), (64, 65), (256, 256), (135, 240)])
@pytest.mark.parametrize("smooth", [False, True])
def test_hip_lf_dequant(ctx, orc, shape, smooth):
    rng = np.random.default_rng(shape[0] * 300 + shape[1] + smooth)
    q = rng.integers(-2000, 2000, size=(3,) + shape).astype(np.int32)
    args = dict(extra_precision=int(rng.integers(0, 4)), x_factor_lf=int(rng.integers(0, 256)), b_factor_lf=int(rng.integers(0, 256)))
    got = host.LFCoefficients.dequantLFCoeff(ctx, q, SD, args["extra_precision"], args["x_factor_lf"], args["b_factor_lf"], smooth,
                                             0.0, 1.0, 84)
    exp = orc.lf_dequant(q, SD, adaptive_smoothing=smooth, **args)
    assert_bits_equal(got, exp, "lf %s" % (shape,))


@pytest.mark.gpu
def test_frame_with_integer_lf_matches_float_lf_path(ctx, orc):
    """whole frame fed with the INTEGER LF image (device runs the LF stage) == frame fed with the oracle's dequantised LF"""
    frame = synth.make_vardct_frame(2304, 264, seed=17, mix="default")  # two LF groups
    rng = np.random.default_rng(5)
    lfq = []
    for g in frame["lfgroups"]:
        h, w = g["dct_select"].shape
        q = rng.integers(-400, 400, size=(3, h, w)).astype(np.int32)
        lfq.append(q)
        lf = orc.lf_dequant(q, SD, extra_precision=1, x_factor_lf=120, b_factor_lf=131, adaptive_smoothing=True,
                            base_corr_x=frame["params"].base_corr_x, base_corr_b=frame["params"].base_corr_b,
                            color_factor=frame["params"].color_factor)
        g["lf"] = [np.ascontiguousarray(lf[c]) for c in range(3)]
    exp = orc.vardct_frame(frame)
    p = abi.VarDCTParams.from_buffer_copy(frame["params"])
    fr = host.Frame(ctx, p, frame["weights"], frame["woffs"])
    for g, q in zip(frame["lfgroups"], lfq):
        fr.setLFGroup(dict(g, lf=None))
        fr.setLFGroupQuant(g["lfg_y"], g["lfg_x"], q, SD, 1, 120, 131, True)
    for grp in range(synth.num_groups(frame)):
        fr.putGroup(0, grp, synth.group_view(frame, grp))
    assert_bits_equal(fr.decodeFrame(), exp, "integer-LF frame")

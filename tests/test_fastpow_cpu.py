"""The transfer stage's pow (csrc/jxl_fastpow.h) restated in numpy with the header's own coefficients: its accuracy
against 80-bit long double, and the float results of the whole PQ / sRGB curves against libm's pow (CPU-only)."""
import math
import os
import re

import numpy as np

HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jxlatte_amd", "csrc", "jxl_fastpow.h")


def _coeffs():
    src = open(HDR).read()
    body = src[src.index("fp_pow_pos"):src.index("fp_pow(double x, double p)")]
    p_first = float(re.search(r"double P = ([0-9.e+-]+);", body).group(1))
    q_first = float(re.search(r"double Q = ([0-9.e+-]+);", body).group(1))
    p_rest = [float(v) for v in re.findall(r"P = __builtin_fma\(P, t2, ([0-9.e+-]+)\);", body)]
    q_rest = [float(v) for v in re.findall(r"Q = __builtin_fma\(Q, rr, ([0-9.e+-]+)\);", body)]
    return [p_first] + p_rest, [q_first] + q_rest  # highest order first


def fast_pow(x, p):
    cl, ce = _coeffs()
    x = np.asarray(x, np.float64)
    m, e = np.frexp(x)
    small = m < 0.70710678118654757
    m = np.where(small, m * 2, m)
    e = np.where(small, e - 1, e)
    num, den = m - 1.0, m + 1.0
    r = 1.0 / den
    t = num * r
    t = t + (num - den * t) * r
    t2 = t * t
    P = np.full_like(t, cl[0])
    for c in cl[1:]:
        P = P * t2 + c
    z = p * (e + t * P)
    n = np.rint(z)
    rr = z - n
    Q = np.full_like(rr, ce[0])
    for c in ce[1:]:
        Q = Q * rr + c
    return np.ldexp(Q, n.astype(np.int64))


def test_header_coefficients_are_the_series():
    cl, ce = _coeffs()
    ln2 = math.log(2.0)
    assert len(cl) == 11 and len(ce) == 14
    for k, c in enumerate(reversed(cl)):
        assert c == 2.0 / (ln2 * (2 * k + 1))           # (2 / ln 2) / (2k + 1): atanh series of log2
    for k, c in enumerate(reversed(ce)):
        assert abs(c - ln2 ** k / math.factorial(k)) <= 1e-17 * max(1.0, c)  # Taylor series of 2^r


def test_accuracy_against_long_double():
    rng = np.random.default_rng(0)
    xs = np.concatenate([rng.uniform(0, 1, 100000), 10 ** rng.uniform(-38, 4, 100000), rng.uniform(0.83, 1.01, 100000)])
    xs = xs.astype(np.float32).astype(np.float64)
    xs = xs[xs > 0]
    for p, tol in ((0.159423828125, 1e-14), (78.84375, 5e-13), (0.4166666666666667, 2e-14)):
        with np.errstate(over="ignore"):
            ref = np.power(xs.astype(np.longdouble), np.longdouble(p))
            got = fast_pow(xs, p)
        with np.errstate(over="ignore"):
            r64 = ref.astype(np.float64)
        ok = np.isfinite(r64) & (r64 > 1e-300)
        rel = np.abs((got[ok].astype(np.longdouble) - ref[ok]) / ref[ok])
        assert float(rel.max()) < tol, (p, float(rel.max()))


def test_pq_and_srgb_float_results_equal_libm():
    rng = np.random.default_rng(1)
    f = np.concatenate([rng.uniform(0, 1, 400000), rng.uniform(0, 12, 40000), 10 ** rng.uniform(-30, 0, 40000)]).astype(np.float32)

    def pq(powf):
        d = powf(f.astype(np.float64), 0.159423828125)
        return powf((0.8359375 + 18.8515625 * d) / (1.0 + 18.6875 * d), 78.84375).astype(np.float32)

    a, b = pq(np.power), pq(fast_pow)
    assert (a.view(np.uint32) != b.view(np.uint32)).mean() < 1e-5
    g = f[f >= 0.0031307]
    s1 = np.power(g.astype(np.float64), 0.4166666666666667).astype(np.float32)
    s2 = fast_pow(g.astype(np.float64), 0.4166666666666667).astype(np.float32)
    assert (s1.view(np.uint32) != s2.view(np.uint32)).mean() < 1e-5


def _pq_table():
    import ctypes as C
    from jxlatte_amd import _lib
    lib = _lib.load()
    n = (129 - 87) * 128
    out = np.zeros((n, 4), np.float32)
    lib.jxl_debug_pq_table.restype = None
    lib.jxl_debug_pq_table.argtypes = [C.c_void_p]
    lib.jxl_debug_pq_table(out.ctypes.data_as(C.c_void_p))
    return out


def _pq_tab_eval(x, tab):
    """fp_tf_pq_tab of csrc/jxl_fastpow.h restated with float32 roundings (each fused multiply-add = one rounding of the
    float64 value of t * a + b, which is exact to 2^-53)"""
    b = x.view(np.uint32)
    idx = (b >> np.uint32(16)).astype(np.int64) - (87 << 7)
    assert ((idx >= 0) & (idx < len(tab))).all()
    sg = tab[idx]
    xm = ((b & np.uint32(0xFFFF0000)) | np.uint32(0x00008000)).view(np.float32)
    t = (x - xm).astype(np.float32)
    assert np.array_equal(t.astype(np.float64), x.astype(np.float64) - xm.astype(np.float64))  # exact
    inner = (t.astype(np.float64) * sg[:, 3].astype(np.float64) + sg[:, 2].astype(np.float64)).astype(np.float32)
    corr = (t.astype(np.float64) * inner.astype(np.float64) + sg[:, 1].astype(np.float64)).astype(np.float32)
    return (sg[:, 0].astype(np.float64) + corr.astype(np.float64)).astype(np.float32), corr, sg


def _pq_ld(x):
    x = x.astype(np.longdouble)
    d = np.power(x, np.longdouble(0.159423828125))
    return np.power((np.longdouble(0.8359375) + np.longdouble(18.8515625) * d) / (np.longdouble(1.0) + np.longdouble(18.6875) * d),
                    np.longdouble(78.84375))


def test_pq_table_within_one_ulp_of_the_reference_form():
    """the tabulated PQ (the device's fast path) on every segment: both ends, the midpoint neighbours and random members:
    within 0.53 ulp of the long double value, hence within 1 ulp of the reference's (float) of a double pow; the GPU sweep
    over all 2^32 inputs is tools/pq_sweep.py"""
    tab = _pq_table()
    rng = np.random.default_rng(3)
    n = len(tab)
    base = ((np.arange(n, dtype=np.uint32) + np.uint32(87 << 7)) << np.uint32(16))
    offs = np.concatenate([np.array([0, 1, 0x7FFF, 0x8000, 0x8001, 0xFFFE, 0xFFFF], np.uint32),
                           rng.integers(0, 1 << 16, 57, dtype=np.uint32)])
    bits = (base[:, None] | offs[None, :]).reshape(-1)
    x = bits.view(np.float32)
    got, corr, sg = _pq_tab_eval(x, tab)
    exact = _pq_ld(x)
    ulp = np.spacing(np.abs(exact.astype(np.float32))).astype(np.longdouble)
    err = np.abs(got.astype(np.longdouble) - exact) / ulp
    assert float(err.max()) <= 0.53, float(err.max())
    # the correction term is small against a0: its own rounding cannot matter
    assert float(np.abs(corr / sg[:, 0]).max()) < 0.02
    # against the reference's form (double pow, cast to float)
    d = np.power(x.astype(np.float64), 0.159423828125)
    ref = np.power((0.8359375 + 18.8515625 * d) / (1.0 + 18.6875 * d), 78.84375).astype(np.float32)
    du = np.abs(got.view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
    assert int(du.max()) <= 1 and float((du != 0).mean()) < 0.02


def test_pq_of_zero_constant_in_the_header():
    """fp_tf_pq_tab returns PQ(+-0) = (float)pow(0.8359375, 78.84375) as a literal (the common out-of-table input besides negative
    samples): the literal in csrc/jxl_fastpow.h is that float, by libm, by long double and by the oracle's transfer()"""
    import re
    from oracle import pyoracle as orc
    src = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "jxlatte_amd", "csrc", "jxl_fastpow.h")).read()
    lit = int(re.search(r"== 0u\) return __builtin_bit_cast\(float, (0x[0-9A-Fa-f]+)u\)", src).group(1), 16)
    assert lit == int(np.float32(np.power(0.8359375, 78.84375)).view(np.uint32))
    assert lit == int(np.float32(np.power(np.longdouble(0.8359375), np.longdouble(78.84375))).view(np.uint32))
    z = orc.transfer(np.array([0.0, -0.0], np.float32), 1)
    assert int(z[0].view(np.uint32)) == lit and int(z[1].view(np.uint32)) == lit


def _srgb8_lookup(x, tab):
    """numpy restatement of fp_srgb8 (csrc/jxl_fastpow.h) on the table build_srgb8_table makes"""
    b = x.view(np.uint32)
    idx = (b >> np.uint32(16)).astype(np.int64) - (118 << 7)
    inside = (idx >= 0) & (idx < 9 * 128)
    q = np.zeros(x.shape, np.int64)
    sg = tab[np.where(inside, idx, 0)]
    with np.errstate(invalid="ignore", over="ignore"):
        qi = sg[:, 0].astype(np.int64) + (x >= sg[:, 1]) + (x >= sg[:, 2]) + (x >= sg[:, 3])
        v = (x * np.float32(12.92)) * np.float32(255.0) + np.float32(0.5)
        lin = np.where(v >= 255.0, 255, np.where(v >= 0.0, np.nan_to_num(v, nan=0.0, posinf=0.0, neginf=0.0).astype(np.int64), 0))
    q = np.where(inside, qi, np.where(x >= 1.0, 255, np.where(np.isnan(x), 0, lin)))
    return q


def test_srgb8_threshold_table_equals_the_reference_composite():
    """sRGB + castToIntWithMax(255) as thresholds: the table the library builds, looked up as the device does, gives the oracle's
    integer for every threshold's neighbourhood, the special values and 4 million random floats (all 2^32 inputs: tools/pq_sweep.py
    --srgb8 on the GPU box, profiles/r3_srgb8_sweep.txt)"""
    import ctypes as C
    from jxlatte_amd import _lib
    from oracle import pyoracle as orc
    lib = _lib.load()
    tab = np.zeros(9 * 128 * 4, np.float32)
    lib.jxl_debug_srgb8_table.restype = C.c_int
    lib.jxl_debug_srgb8_table.argtypes = [C.c_void_p]
    assert lib.jxl_debug_srgb8_table(tab.ctypes.data) == 0
    tab = tab.reshape(-1, 4)
    assert tab[0, 0] >= 6 and tab[-1, 0] <= 255 and np.all(np.diff(tab[:, 0]) >= 0)
    thr = tab[:, 1:][np.isfinite(tab[:, 1:])]
    assert thr.size == int(tab[-1, 0] + np.isfinite(tab[-1, 1:]).sum() - tab[0, 0])  # one threshold per level step of [2^-9, 1)
    tb = thr.view(np.uint32).astype(np.int64)
    near = np.concatenate([tb + d for d in (-2, -1, 0, 1, 2)]).astype(np.uint32).view(np.float32)
    rng = np.random.default_rng(5)
    special = np.array([0.0, -0.0, 1e-45, -1e-45, 2.0 ** -9, np.nextafter(np.float32(2.0 ** -9), np.float32(0)), 0.0031306684, 0.00313066844250063,
                        0.0031306685, 0.5, 0.99999994, 1.0, 1.0000001, 2.0, 1e30, np.inf, -np.inf, np.nan, -1.0, -1e-4, 1e-4, 1.95e-3], np.float32)
    x = np.concatenate([near, special, rng.random(2000000).astype(np.float32), (10.0 ** rng.uniform(-12, 1, 1000000)).astype(np.float32),
                        -(10.0 ** rng.uniform(-12, 1, 200000)).astype(np.float32),
                        rng.integers(0, 2 ** 32, 800000, dtype=np.uint64).astype(np.uint32).view(np.float32)])
    exp = orc.transfer(x, 2, 255)
    got = _srgb8_lookup(x, tab)
    bad = np.flatnonzero(got != exp)
    assert bad.size == 0, (x[bad[:5]], got[bad[:5]], exp[bad[:5]])


def test_pq16_thresholds_describe_the_reference_composite():
    """PQ + castToIntWithMax(65535) as thresholds (build_pq16_thresholds): strictly increasing, and for every level k the threshold
    is the first float that reaches it -- thr[k] gives k, the float just below gives k - 1 (oracle's transfer())"""
    import ctypes as C
    from jxlatte_amd import _lib
    from oracle import pyoracle as orc
    lib = _lib.load()
    thr = np.zeros(65537, np.float32)
    lib.jxl_debug_pq16_thresholds.restype = None
    lib.jxl_debug_pq16_thresholds.argtypes = [C.c_void_p]
    lib.jxl_debug_pq16_thresholds(thr.ctypes.data)
    assert thr[0] == -np.inf and thr[65536] == np.inf
    t = thr[1:65536]
    assert np.all(np.diff(t.view(np.uint32).astype(np.int64)) > 0) and t[0] > 0 and t[-1] <= 1.0
    assert np.array_equal(orc.transfer(t, 1, 65535), np.arange(1, 65536))
    below = (t.view(np.uint32) - np.uint32(1)).view(np.float32)
    assert np.array_equal(orc.transfer(below, 1, 65535), np.arange(0, 65535))


def test_srgb16_thresholds_describe_the_reference_composite():
    """sRGB + castToIntWithMax(65535) as thresholds (build_srgb16_table): strictly increasing; thr[k] gives level k, the float just
    below gives k - 1 (oracle's transfer()); and the segment table in front of them is the curve to float accuracy"""
    import ctypes as C
    from jxlatte_amd import _lib
    from oracle import pyoracle as orc
    lib = _lib.load()
    nseg = 9 * 128 * 4
    buf = np.zeros(nseg + 65537, np.float32)
    lib.jxl_debug_srgb16_table.restype = None
    lib.jxl_debug_srgb16_table.argtypes = [C.c_void_p]
    lib.jxl_debug_srgb16_table(buf.ctypes.data)
    thr = buf[nseg:]
    assert thr[0] == -np.inf and thr[65536] == np.inf
    t = thr[1:65536]
    assert np.all(np.diff(t.view(np.uint32).astype(np.int64)) > 0) and t[0] > 0 and t[-1] <= 1.0
    assert np.array_equal(orc.transfer(t, 2, 65535), np.arange(1, 65536))
    below = (t.view(np.uint32) - np.uint32(1)).view(np.float32)
    assert np.array_equal(orc.transfer(below, 2, 65535), np.arange(0, 65535))
    # the segments: value at each segment midpoint against the oracle's float curve (<= 2 ulp: three roundings against one)
    seg = buf[:nseg].reshape(-1, 4)
    mid = ((np.arange(9 * 128, dtype=np.uint32) + np.uint32(118 << 7)) << np.uint32(16) | np.uint32(0x8000)).view(np.float32)
    ok = mid >= np.float32(0.0031306685)
    ref = orc.transfer(mid[ok], 2)
    d = np.abs(seg[ok, 0].view(np.int32).astype(np.int64) - ref.view(np.int32).astype(np.int64))
    assert d.max() <= 2, d.max()

"""Rows f4 / f3 of the scope table: chroma upsampling, k-times upsampling, noise synthesis, blending, orientation and
sample packing. CPU tests pin the oracle with analytic known answers and numpy restatements; GPU tests compare the
HIP kernels with the oracle bit for bit through the C-ABI (host mirror functions of jxlatte_amd.host)."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from jxlatte_amd import abi


def rnd_plane(rng, h, w, specials=True):
    a = rng.standard_normal((h, w)).astype(np.float32)
    if specials and a.size >= 8:
        flat = a.reshape(-1)
        idx = rng.choice(a.size, size=min(6, a.size), replace=False)
        for i, v in zip(idx, (np.nan, np.inf, -np.inf, -0.0, 1e-42, 3e38)):
            flat[i] = np.float32(v)
    return a


# ---- oracle known answers (CPU) ----------------------------------------------------------------------------
def test_chroma_upsample_kats(orc):
    c = np.full((5, 7), 2.5, np.float32)
    assert_bits_equal(orc.chroma_upsample(c, 1, 1), np.full((10, 14), 2.5, np.float32))
    row = np.array([[0, 4, 8]], np.float32)
    # Frame.java:694-696: 0.75*x + 0.25*left | right, edges replicated
    assert_bits_equal(orc.chroma_upsample(row, 1, 0), np.array([[0, 1, 3, 5, 7, 8]], np.float32))
    col = row.T.copy()
    assert_bits_equal(orc.chroma_upsample(col, 0, 1), np.array([[0, 1, 3, 5, 7, 8]], np.float32).T.copy())
    # two doublings = doubling twice
    a = rnd_plane(np.random.default_rng(1), 6, 5, specials=False)
    assert_bits_equal(orc.chroma_upsample(a, 2, 0), orc.chroma_upsample(orc.chroma_upsample(a, 1, 0), 1, 0))
    # horizontal passes run before vertical ones (:683-721)
    assert_bits_equal(orc.chroma_upsample(a, 1, 1), orc.chroma_upsample(orc.chroma_upsample(a, 1, 0), 0, 1))


@pytest.mark.parametrize("k", [2, 4, 8])
def test_upsampling_weight_expansion(orc, k):
    n = {2: 15, 4: 55, 8: 210}[k]
    packed = np.arange(n, dtype=np.float32)
    w = orc.upsampling_weights(k, packed)
    # the full (5k/2)^2 table is symmetric and the four quadrants mirror each other (ImageHeader.java:458-459)
    assert_bits_equal(w, np.ascontiguousarray(w.transpose(1, 0, 3, 2)))
    assert_bits_equal(w, np.ascontiguousarray(w[::-1, :, ::-1, :]))
    assert_bits_equal(w, np.ascontiguousarray(w[:, ::-1, :, ::-1]))
    # every packed coefficient is used, the first is the (0,0) corner
    assert set(np.unique(w).astype(int)) == set(range(n))
    assert w[0, 0, 0, 0] == 0 and w[0, 0, 0, 1] == 1
    # the product library's host helper computes the same table
    from jxlatte_amd import host
    assert_bits_equal(host.getUpWeights(k, packed), w)


@pytest.mark.parametrize("k", [2, 4, 8])
def test_upsample_kats(orc, k):
    rng = np.random.default_rng(k)
    a = np.abs(rnd_plane(rng, 9, 6, specials=False)) + 0.5
    delta = np.zeros((k, k, 5, 5), np.float32)
    delta[:, :, 2, 2] = 1.0
    # centre-only kernel = nearest-neighbour replication
    assert_bits_equal(orc.upsample(a, k, delta), np.repeat(np.repeat(a, k, 0), k, 1))
    # results are clamped into the 5x5 window's [min, max]
    big = np.full((k, k, 5, 5), 10.0, np.float32)
    up = orc.upsample(a, k, big)
    pad = np.pad(a, 2, mode="symmetric")
    win_max = np.max([pad[i:i + 9, j:j + 6] for i in range(5) for j in range(5)], axis=0)
    assert_bits_equal(up, np.repeat(np.repeat(win_max, k, 0), k, 1))
    # the reference starts max at Float.MIN_VALUE (Frame.java:237): an all-negative window clamps to 1.4e-45 from above
    neg = -a
    up = orc.upsample(neg, k, -big)
    assert np.all(up == np.float32(1.4e-45))


def test_noise_kats(orc):
    n = orc.noise_init(40, 300, seed0=(7 << 32) | 3, group_dim=256)
    assert n.shape == (3, 40, 300) and np.isfinite(n).all()
    # local samples are in [1,2): the high-pass of them lies inside (-3.84, 3.84) and is zero-mean
    assert np.abs(n).max() < 3.84 and abs(float(n.mean())) < 0.02
    assert 0.9 < float(n.std()) < 1.3  # sqrt(24*0.16^2 + 3.84^2) / sqrt(12) = 1.13
    assert_bits_equal(n, orc.noise_init(40, 300, seed0=(7 << 32) | 3, group_dim=256))
    assert not np.array_equal(n, orc.noise_init(40, 300, seed0=(7 << 32) | 4, group_dim=256))
    # group streams depend on the group origin only: the interior of group (0,0) is the same in a wider image
    m = orc.noise_init(40, 600, seed0=(7 << 32) | 3, group_dim=256)
    assert_bits_equal(n[:, 2:38, 2:254], m[:, 2:38, 2:254])
    # splitMix64 known answer (Vigna's reference implementation, seed 0 first output; XorShiro.java:9-13)
    import ctypes as C
    lut0 = np.zeros(8, np.float32)
    p = rnd_plane(np.random.default_rng(2), 3, 10 * 40, specials=False).reshape(3, 10, 40)
    assert_bits_equal(orc.noise_add(p, n[:, :10, :40], lut0, 0.0, 1.0), p)  # strength 0: unchanged


def test_xorshiro_stream_python_restatement(orc):
    """independent pure-Python restatement of XorShiro + the bit-to-float step for the first row of a group"""
    M = (1 << 64) - 1

    def sm(z):
        z = ((z ^ (z >> 30)) * 0xbf58476d1ce4e5b9) & M
        z = ((z ^ (z >> 27)) * 0x94d049bb133111eb) & M
        return z ^ (z >> 31)
    assert sm(0x9e3779b97f4a7c15) == 0xe220a8397b1dcdaf  # splitmix64(seed 0) first output, published test vector
    seed0, x0, y0 = 0x1234567800000009, 16, 0
    s0 = [sm((seed0 + 0x9e3779b97f4a7c15) & M)]
    s1 = [sm((((x0 << 32) | y0) + 0x9e3779b97f4a7c15) & M)]
    for i in range(1, 8):
        s0.append(sm(s0[-1]))
        s1.append(sm(s1[-1]))
    vals = []
    for _ in range(1):
        for i in range(8):
            a, b = s1[i], s0[i]
            c = (a + b) & M
            s0[i] = a
            b ^= (b << 23) & M
            s1[i] = b ^ a ^ (b >> 18) ^ (a >> 5)
            vals += [c & 0xffffffff, c >> 32]
    local = np.array([(v >> 9) | 0x3f800000 for v in vals], np.uint32).view(np.float32)
    # a 1-row, 2-group image with 16-px groups: the second group's 16 samples are exactly one batch; with h == 1 and
    # mirrored edges the 5x5 sum reads row 0 five times
    n = orc.noise_init(1, 32, seed0, group_dim=16, colors=1)[0, 0]
    first = orc.noise_init(1, 16, seed0, group_dim=16, colors=1)
    assert first.shape == (1, 1, 16)
    # recompute pixel x = 24 (inside the second group, window 22..26 stays inside it)
    acc = np.float32(0)
    for iy in range(5):
        for ix in range(5):
            wgt = np.float32(-3.84) if (iy == 2 and ix == 2) else np.float32(0.16)
            acc = np.float32(acc + np.float32(local[24 - 16 + ix - 2] * wgt))
    assert n[24] == acc


def test_noise_add_formula(orc):
    rng = np.random.default_rng(5)
    p = (rng.random((3, 4, 9)).astype(np.float32) - 0.2)
    nz = rng.standard_normal((3, 4, 9)).astype(np.float32)
    lut = rng.random(8).astype(np.float32) * 1.5
    got = orc.noise_add(p, nz, lut, 0.25, 0.75)
    f = np.float32
    x, y, b = p
    inr = np.where(y + x < 0, f(0), f(3) * (y + x)).astype(f)
    ing = np.where(y - x < 0, f(0), f(3) * (y - x)).astype(f)

    def strength(v):
        i = np.where(v >= 7, 6, v.astype(np.int32))
        fr = np.where(v >= 7, f(1), v - i.astype(f)).astype(f)
        s = ((lut[i + 1] - lut[i]) * fr + lut[i]).astype(f)
        return np.clip(s, f(0), f(1))
    nr = strength(inr) * (f(0.00171875) * nz[0] + f(0.21828125) * nz[2])
    ng = strength(ing) * (f(0.00171875) * nz[1] + f(0.21828125) * nz[2])
    nrg = nr + ng
    exp = np.stack([x + (f(0.25) * nrg + nr - ng), y + nrg, b + f(0.75) * nrg]).astype(f)
    assert_bits_equal(got, exp)


def test_blend_kats(orc):
    rng = np.random.default_rng(3)
    cv = rng.random((8, 10)).astype(np.float32)
    fr = rng.random((6, 7)).astype(np.float32)
    rf = rng.random((8, 10)).astype(np.float32)
    fa = rng.random((6, 7)).astype(np.float32) * 1.4 - 0.2
    ra = rng.random((8, 10)).astype(np.float32)
    rect = (4, 5, 2, 3, 1, 1, 2, 3)
    cs, fs, rs = (slice(2, 6), slice(3, 8)), (slice(1, 5), slice(1, 6)), (slice(2, 6), slice(3, 8))
    f1 = np.float32(1)

    def expect(v):
        e = cv.copy()
        e[cs] = v
        return e
    st, got = orc.blend(abi.BLEND_REPLACE, cv, fr, None, rect)
    assert st == 0
    assert_bits_equal(got, expect(fr[fs]))
    st, got = orc.blend(abi.BLEND_ADD, cv, fr, rf, rect)
    assert_bits_equal(got, expect(rf[rs] + fr[fs]))
    st, got = orc.blend(abi.BLEND_MULT, cv, fa, rf, rect, clamp=True)
    assert_bits_equal(got, expect(np.clip(fa[fs], 0, 1) * rf[rs]))
    # BLEND without extra channels degrades to ADD (:346-349)
    st, got = orc.blend(abi.BLEND_BLEND, cv, fr, rf, rect, frame_alpha=fa, ref_alpha=ra)
    assert_bits_equal(got, expect(rf[rs] + fr[fs]))
    na = np.clip(fa[fs], 0, 1)
    st, got = orc.blend(abi.BLEND_BLEND, cv, fr, rf, rect, frame_alpha=fa, ref_alpha=ra, has_extra=True, clamp=True)
    assert_bits_equal(got, expect((fr[fs] * na + rf[rs] * ra[rs] * (f1 - na)) / (ra[rs] + na * (f1 - ra[rs]))))
    st, got = orc.blend(abi.BLEND_BLEND, cv, fr, rf, rect, frame_alpha=fa, ref_alpha=ra, has_extra=True, clamp=True, premult=True)
    assert_bits_equal(got, expect(fr[fs] + rf[rs] * (f1 - na)))
    st, got = orc.blend(abi.BLEND_BLEND, cv, fr, rf, rect, has_extra=True, is_alpha=True)
    assert_bits_equal(got, expect(rf[rs] + fr[fs] * (f1 - rf[rs])))
    st, got = orc.blend(abi.BLEND_MULADD, cv, fr, rf, rect, frame_alpha=fa, has_extra=True)
    assert_bits_equal(got, expect(rf[rs] + fa[fs] * fr[fs]))
    # the alpha channel under MULADD keeps the old alpha, read at frameOffset (:396-398)
    st, got = orc.blend(abi.BLEND_MULADD, cv, fr, rf, rect, has_extra=True, is_alpha=True)
    assert_bits_equal(got, expect(rf[fs]))
    # ints: add wraps like Java
    ci = np.zeros((2, 2), np.int32)
    st, got = orc.blend(abi.BLEND_ADD, ci, np.full((2, 2), 2**31 - 1, np.int32), np.full((2, 2), 5, np.int32), (2, 2, 0, 0, 0, 0, 0, 0))
    assert st == 0 and np.all(got == -(2**31) + 4)
    st, _ = orc.blend(7, cv, fr, rf, rect)
    assert st == abi.JXL_ERR_INVALID_BITSTREAM
    st, _ = orc.blend(abi.BLEND_MULT, ci, ci, ci, (2, 2, 0, 0, 0, 0, 0, 0))
    assert st == abi.JXL_ERR_INVALID_ARGUMENT


def np_orient(a, o):
    return {1: a, 2: a[:, ::-1], 3: a[::-1, ::-1], 4: a[::-1, :], 5: a.T, 6: a.T[:, ::-1], 7: a[::-1, ::-1].T, 8: a.T[::-1, :]}[o]


@pytest.mark.parametrize("o", range(1, 9))
def test_orient_oracle_vs_numpy(orc, o):
    a = np.arange(5 * 7, dtype=np.int32).reshape(5, 7)
    assert_bits_equal(orc.orient(a, o), np.ascontiguousarray(np_orient(a, o)))
    # EXIF semantics: 6 = rotate 90 clockwise, 8 = counter-clockwise, 3 = 180 degrees
    if o == 6:
        assert_bits_equal(orc.orient(a, o), np.ascontiguousarray(np.rot90(a, -1)))
    if o == 8:
        assert_bits_equal(orc.orient(a, o), np.ascontiguousarray(np.rot90(a, 1)))
    if o == 3:
        assert_bits_equal(orc.orient(a, o), np.ascontiguousarray(np.rot90(a, 2)))


def np_quant(f, maxv):
    with np.errstate(all="ignore"):
        v = f.astype(np.float32) * np.float32(maxv) + np.float32(0.5)
        v = np.where(np.isnan(v), 0, np.clip(np.trunc(np.clip(v, -3e9, 3e9)), -2**31, 2**31 - 1))
    return np.clip(v, 0, maxv).astype(np.int64)


def test_pack_oracle_vs_numpy(orc):
    rng = np.random.default_rng(9)
    planes = [rnd_plane(rng, 6, 11) * 0.5 + 0.5 for _ in range(3)]
    out = orc.pack(planes, 8)
    assert out.dtype == np.uint8 and out.shape == (6, 11, 3)
    for c in range(3):
        assert np.array_equal(out[..., c], np_quant(planes[c], 255))
    out16 = orc.pack(planes, 16)
    be = orc.pack(planes, 16, big_endian=True)
    assert np.array_equal(out16, be.byteswap())
    for c in range(3):
        assert np.array_equal(out16[..., c], np_quant(planes[c], 65535))
    # ints with matching depth are only clamped; a different depth is rescaled through float (PNGWriter.java:79-90)
    ints = [rng.integers(-20, 300, (6, 11)).astype(np.int32) for _ in range(3)]
    o = orc.pack(ints, 8)
    for c in range(3):
        assert np.array_equal(o[..., c], np.clip(ints[c], 0, 255))
    o = orc.pack(ints, 16, tagged_depth=[8, 8, 8])
    f = np.float32
    for c in range(3):
        assert np.array_equal(o[..., c], np_quant(ints[c].astype(f) * (f(1) / f(255)), 65535))
    # premultiplied alpha: colour / alpha, alpha itself quantised as is; gray + alpha layout
    g = rng.random((6, 11)).astype(np.float32)
    al = rng.random((6, 11)).astype(np.float32)
    al[0, 0] = 0
    o = orc.pack([g], 8, alpha=al, premultiplied=True)
    assert o.shape == (6, 11, 2)
    with np.errstate(all="ignore"):
        assert np.array_equal(o[..., 0], np_quant(g / al, 255))
    assert np.array_equal(o[..., 1], np_quant(al, 255))


# ---- HIP vs oracle (GPU) -----------------------------------------------------------------------------------
SHAPES = [(1, 1), (1, 9), (2, 3), (7, 2), (33, 65), (64, 64), (100, 131)]


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_chroma_upsample_gpu(ctx, orc, shape):
    from jxlatte_amd import host
    a = rnd_plane(np.random.default_rng(shape[0] * 131 + shape[1]), *shape)
    for xs, ys in ((1, 0), (0, 1), (1, 1), (2, 1), (0, 0)):
        assert_bits_equal(host.invertSubsampling(ctx, a, xs, ys), orc.chroma_upsample(a, xs, ys), "%s %d %d" % (shape, xs, ys), any_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("k", [2, 4, 8])
@pytest.mark.parametrize("shape", [(1, 1), (2, 5), (3, 2), (37, 50), (64, 96)])
def test_upsample_gpu(ctx, orc, k, shape):
    from jxlatte_amd import host
    rng = np.random.default_rng(k * 1000 + shape[0] * 7 + shape[1])
    a = rnd_plane(rng, *shape)
    n = {2: 15, 4: 55, 8: 210}[k]
    wts = orc.upsampling_weights(k, (rng.standard_normal(n) * 0.2).astype(np.float32))
    assert_bits_equal(host.performUpsampling(ctx, a, k, wts), orc.upsample(a, k, wts), "k=%d %s" % (k, shape), any_nan=True)
    b = -np.abs(rnd_plane(rng, *shape, specials=False)) - 1  # all-negative windows: the Float.MIN_VALUE quirk
    assert_bits_equal(host.performUpsampling(ctx, b, k, wts), orc.upsample(b, k, wts), "neg k=%d %s" % (k, shape), any_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("h,w,gd,colors", [(1, 1, 256, 3), (5, 40, 16, 1), (300, 520, 256, 3), (130, 129, 128, 3), (64, 1000, 512, 2)])
def test_noise_init_gpu(ctx, orc, h, w, gd, colors):
    from jxlatte_amd import host
    seed = (3 << 32) | (h * w)
    assert_bits_equal(host.initializeNoise(ctx, h, w, seed, gd, colors), orc.noise_init(h, w, seed, gd, colors), "noise %dx%d" % (h, w), any_nan=True)


@pytest.mark.gpu
def test_noise_add_gpu(ctx, orc):
    from jxlatte_amd import host
    rng = np.random.default_rng(77)
    p = np.stack([rnd_plane(rng, 50, 67) for _ in range(3)])
    p[1] += 1.0
    p[:, 10:20] *= 4  # drive the >= 7 branch
    nz = orc.noise_init(50, 67, 99)
    lut = (rng.random(8) * 1.6 - 0.3).astype(np.float32)
    assert_bits_equal(host.synthesizeNoise(ctx, p, nz, lut, 0.0, 1.0), orc.noise_add(p, nz, lut, 0.0, 1.0), any_nan=True)
    assert_bits_equal(host.synthesizeNoise(ctx, p, nz, lut, -0.3, 0.935), orc.noise_add(p, nz, lut, -0.3, 0.935), any_nan=True)


@pytest.mark.gpu
def test_blend_gpu(ctx, orc):
    from jxlatte_amd import host, _lib
    rng = np.random.default_rng(21)
    cv, rf, ra = (rnd_plane(rng, 40, 70) for _ in range(3))
    fr, fa = (rnd_plane(rng, 33, 90) for _ in range(2))
    # frame offsets also fit the reference plane: blendMulAdd's alpha case reads ref at frameOffset
    rects = [(20, 50, 5, 7, 3, 12, 10, 2), (1, 1, 39, 69, 32, 69, 0, 0), (33, 70, 7, 0, 0, 0, 0, 0), (0, 0, 0, 0, 0, 0, 0, 0)]
    for rect in rects:
        for mode in range(5):
            for flags in range(16):
                kw = dict(isAlpha=bool(flags & 1), hasExtra=bool(flags & 2), clamp=bool(flags & 4), premult=bool(flags & 8))
                okw = dict(is_alpha=kw["isAlpha"], has_extra=kw["hasExtra"], clamp=kw["clamp"], premult=kw["premult"])
                st, exp = orc.blend(mode, cv, fr, rf, rect, frame_alpha=fa, ref_alpha=ra, **okw)
                assert st == 0
                got = host.blend(ctx, mode, cv, fr, rf, rect, frameAlpha=fa, refAlpha=ra, **kw)
                assert_bits_equal(got, exp, "mode %d flags %d rect %s" % (mode, flags, rect), any_nan=True)
    ci, fi, ri = (rng.integers(-2**31, 2**31, s, dtype=np.int64).astype(np.int32) for s in ((40, 70), (33, 90), (40, 70)))
    for mode in (abi.BLEND_REPLACE, abi.BLEND_ADD, abi.BLEND_BLEND, abi.BLEND_MULADD):
        st, exp = orc.blend(mode, ci, fi, ri, rects[0])
        assert st == 0
        assert_bits_equal(host.blend(ctx, mode, ci, fi, ri, rects[0]), exp, "int mode %d" % mode)
    with pytest.raises(_lib.InvalidBitstreamException):
        host.blend(ctx, 9, cv, fr, rf, rects[0])
    with pytest.raises(_lib.IllegalArgumentException):
        host.blend(ctx, abi.BLEND_MULT, ci, fi, ri, rects[0])
    with pytest.raises(_lib.IllegalArgumentException):  # rectangle leaves the frame
        host.blend(ctx, abi.BLEND_ADD, cv, fr, rf, (20, 50, 5, 7, 20, 60, 0, 0))


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(1, 1), (1, 40), (33, 1), (31, 33), (64, 96), (100, 257)])
def test_orient_gpu(ctx, orc, shape):
    from jxlatte_amd import host, _lib
    rng = np.random.default_rng(shape[0] + shape[1])
    a = rng.integers(-2**31, 2**31, shape, dtype=np.int64).astype(np.int32)
    f = rnd_plane(rng, *shape)
    for o in range(1, 9):
        assert_bits_equal(host.transposeBuffer(ctx, a, o), orc.orient(a, o), "int o=%d" % o)
        assert_bits_equal(host.transposeBuffer(ctx, f, o), orc.orient(f, o), "float o=%d" % o)
    with pytest.raises(_lib.IllegalStateException):
        host.transposeBuffer(ctx, a, 9)


@pytest.mark.gpu
def test_pack_gpu(ctx, orc):
    from jxlatte_amd import host, _lib
    rng = np.random.default_rng(31)
    h, w = 45, 83
    fl = [rnd_plane(rng, h, w) * 0.5 + 0.5 for _ in range(3)]
    it = [rng.integers(-1000, 70000, (h, w)).astype(np.int32) for _ in range(3)]
    al_f = rnd_plane(rng, h, w) * 0.5 + 0.5
    al_i = rng.integers(0, 256, (h, w)).astype(np.int32)
    for depth in (8, 16):
        for be in (False, True):
            for planes in (fl, it, [fl[0], it[1], fl[2]], fl[:1], it[:1]):
                for alpha, premult in ((None, False), (al_f, False), (al_f, True), (al_i, True), (al_i, False)):
                    for tagged in (None, [8, 8, 8, 8], [12, 16, 10, 8]):
                        kw = dict(alpha=alpha, premultiplied=premult, big_endian=be)
                        exp = orc.pack(planes, depth, tagged_depth=tagged, **kw)
                        got = host.packSamples(ctx, planes, depth, alpha=alpha, premultiplied=premult, taggedDepth=tagged, bigEndian=be)
                        assert_bits_equal(got, exp, "depth %d be %s premult %s tagged %s" % (depth, be, premult, tagged), any_nan=True)
    with pytest.raises(_lib.IllegalArgumentException):
        host.packSamples(ctx, fl, 12)
    with pytest.raises(_lib.IllegalArgumentException):
        host.packSamples(ctx, fl, 8, premultiplied=True)


@pytest.mark.gpu
@pytest.mark.parametrize("fmt,tf", [(abi.OUT_RGB8, abi.TRANSFER_SRGB), (abi.OUT_RGB16, abi.TRANSFER_PQ), (abi.OUT_RGB16, abi.TRANSFER_NONE)])
@pytest.mark.parametrize("size", [(96, 64), (72, 40)])
def test_frame_interleaved_output(ctx, orc, fmt, tf, size):
    """row f3 inside the frame pipeline: JXL_OUT_RGB8 / RGB16 = the planar quantised samples, pixel-interleaved"""
    from jxlatte_amd import host, synth
    fr = synth.make_vardct_frame(size[0], size[1], seed=size[0] + fmt, aligned=False)
    fr["params"].transfer, fr["params"].out_format = tf, fmt
    got = host.Frame.from_synth(ctx, fr, stages=31).decodeFrame()
    planar_fmt = abi.OUT_U8 if fmt == abi.OUT_RGB8 else abi.OUT_U16
    fr["params"].out_format = planar_fmt
    planar = host.Frame.from_synth(ctx, fr, stages=31).decodeFrame()
    assert got.shape == (size[1], size[0], 3) and got.dtype == planar.dtype
    assert np.array_equal(got, np.moveaxis(planar, 0, -1))
    exp = orc.vardct_frame(fr, stages=31)
    diff = np.abs(got.astype(np.int64) - np.moveaxis(exp, 0, -1).astype(np.int64))
    assert diff.max() <= 1  # transfer stage: <= 1 ulp of the float, i.e. at most one quantisation step
    if tf == abi.TRANSFER_NONE:
        assert diff.max() == 0
    # the small-frame fallback (stage kernels) writes the same layout
    small = synth.make_vardct_frame(8, 8, seed=3)
    small["params"].transfer, small["params"].out_format = tf, fmt
    g2 = host.Frame.from_synth(ctx, small, stages=31).decodeFrame()
    small["params"].out_format = planar_fmt
    assert np.array_equal(g2, np.moveaxis(host.Frame.from_synth(ctx, small, stages=31).decodeFrame(), 0, -1))


# ---- row f4 chained on the device: host.ResidentPlanes ----------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("k", [1, 2, 4, 8])
@pytest.mark.parametrize("size", [(96, 64), (72, 40)])
def test_resident_planes_chain_equals_stage_by_stage(ctx, orc, k, size):
    """a VarDCT frame's planes kept on the device through upsample -> noise -> XYB (JXLCodestreamDecoder.java:628-637) equal
    the oracle applying the same stages to the frame's result, bit for bit; the window is the unpadded frame size"""
    from jxlatte_amd import host, synth
    fr = synth.make_vardct_frame(size[0], size[1], seed=size[0] * 3 + k, aligned=False)
    stages = abi.STAGE_IDCT | abi.STAGE_GAB | abi.STAGE_EPF
    h, w = size[1] - 3, size[0] - 5  # Frame bounds smaller than the padded size
    rng = np.random.default_rng(k + size[0])
    p = fr["params"]
    m = [p.opsin_matrix[i] for i in range(9)]
    ob = [p.opsin_bias[i] for i in range(3)]
    cb = [p.cbrt_opsin_bias[i] for i in range(3)]
    lut = (rng.random(8) * 0.2).astype(np.float32)
    seed = (5 << 32) | 3
    rp = host.Frame.from_synth(ctx, fr, stages=stages).keepPlanes(h, w)
    assert rp.shape == (h, w)
    exp = np.ascontiguousarray(orc.vardct_frame(fr, stages=stages)[:, :h, :w])
    assert_bits_equal(rp.download(), exp, "window", any_nan=True)
    if k > 1:
        n = {2: 15, 4: 55, 8: 210}[k]
        wts = orc.upsampling_weights(k, (rng.standard_normal(n) * 0.2).astype(np.float32))
        rp.upsample(k, wts)
        exp = np.stack([orc.upsample(np.ascontiguousarray(exp[c]), k, wts) for c in range(3)])
        assert rp.shape == (h * k, w * k)
    rp.noise(256, seed, lut, p.base_corr_x, p.base_corr_b)
    exp = orc.noise_add(exp, orc.noise_init(h * k, w * k, seed, 256, 3), lut, p.base_corr_x, p.base_corr_b)
    # the host hook: down, a host-side edit (stands for patches / splines), up again
    mid = rp.download()
    assert_bits_equal(mid, exp, "after noise", any_nan=True)
    mid[:, ::7, ::5] += np.float32(0.125)
    exp[:, ::7, ::5] += np.float32(0.125)
    rp.replace(mid)
    rp.invertXYB(m, ob, cb, p.intensity_target)
    exp = orc.xyb(exp, m, ob, cb, p.intensity_target)
    assert_bits_equal(rp.download(), exp, "after XYB", any_nan=True)
    rp.ycbcr()
    assert_bits_equal(rp.download(), orc.ycbcr(exp), "after YCbCr", any_nan=True)


@pytest.mark.gpu
def test_resident_planes_errors(ctx):
    from jxlatte_amd import _lib, host, synth
    c2 = _lib.Context(0)
    try:
        with pytest.raises(_lib.JxlError):
            host.ResidentPlanes(c2).download()  # nothing resident
        with pytest.raises(_lib.JxlError):
            c2.call("jxl_planes_from_frame", 8, 8)  # nothing run
        fr = synth.make_vardct_frame(64, 64, seed=1)
        fr["params"].transfer, fr["params"].out_format = abi.TRANSFER_SRGB, abi.OUT_U8
        f = host.Frame.from_synth(c2, fr, stages=31)
        with pytest.raises(_lib.JxlError):
            f.keepPlanes(64, 64)  # integer result: not a set of float planes
        fr["params"].transfer, fr["params"].out_format = abi.TRANSFER_NONE, abi.OUT_F32
        f = host.Frame.from_synth(c2, fr, stages=7)
        with pytest.raises(_lib.JxlError):
            f.keepPlanes(65, 64)  # window outside the frame
        rp = f.keepPlanes(64, 64)
        with pytest.raises(_lib.JxlError):
            rp.ctx.call("jxl_planes_upsample", 3, None)
    finally:
        c2.close()

"""GPU parity of the Modular path (inverse Squeeze steps, RCT, int->float): bit-exact vs the oracle."""
import numpy as np
import pytest

from conftest import assert_bits_equal
from jxlatte_amd import _lib, host, synth

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,aw,rw", [(1, 1, 0), (1, 1, 1), (3, 2, 1), (64, 64, 64), (65, 33, 32), (200, 129, 129), (7, 500, 499), (1080, 960, 960)])
def test_inv_hsqueeze(ctx, orc, h, aw, rw):
    rng = np.random.default_rng(h * 3 + aw)
    avg = rng.integers(-3000, 3000, size=(h, aw)).astype(np.int32)
    res = np.rint(rng.laplace(0, 40, size=(h, rw))).astype(np.int32)
    assert_bits_equal(host.ModularChannel.inverseHorizontalSqueeze(ctx, avg, res), orc.inv_hsqueeze(avg, res), "hsq")


@pytest.mark.parametrize("w,ah,rh", [(1, 1, 0), (1, 1, 1), (3, 2, 1), (64, 64, 64), (65, 33, 32), (200, 129, 129), (500, 7, 6), (1920, 540, 540)])
def test_inv_vsqueeze(ctx, orc, w, ah, rh):
    rng = np.random.default_rng(w * 3 + ah)
    avg = rng.integers(-3000, 3000, size=(ah, w)).astype(np.int32)
    res = np.rint(rng.laplace(0, 40, size=(rh, w))).astype(np.int32)
    assert_bits_equal(host.ModularChannel.inverseVerticalSqueeze(ctx, avg, res), orc.inv_vsqueeze(avg, res), "vsq")


def test_squeeze_wraparound_and_tendency_branches(ctx, orc):
    big = np.array([[2 ** 31 - 1, -2 ** 31, 2 ** 31 - 5, 17, -2 ** 31 + 3, 0, 2 ** 30, -2 ** 30]], np.int32)
    res = np.array([[2 ** 31 - 1, -2 ** 31, 12345, -98765, 2 ** 30, -7, 3, 2 ** 31 - 1]], np.int32)
    assert_bits_equal(host.ModularChannel.inverseHorizontalSqueeze(ctx, big, res), orc.inv_hsqueeze(big, res), "wrap h")
    assert_bits_equal(host.ModularChannel.inverseVerticalSqueeze(ctx, big.T.copy(), res.T.copy()), orc.inv_vsqueeze(big.T.copy(), res.T.copy()), "wrap v")
    ramp = np.arange(0, 640, 10, dtype=np.int32).reshape(1, 64)
    for a in (ramp, -ramp, ramp[:, ::-1].copy()):
        r = np.zeros_like(a)
        assert_bits_equal(host.ModularChannel.inverseHorizontalSqueeze(ctx, a, r), orc.inv_hsqueeze(a, r), "ramp")


def test_squeeze_overflowing_differences(ctx, orc):
    """(left, avg, next) triples whose true differences exceed the int32 range -- (INT_MAX, INT_MIN, INT_MIN), alternating
    extremes, extremes next to small values: the wrapped difference lands inside the fast path's "safe" window, so the
    overflow itself has to send the lane to the exact form (k_modular.hip tend_fast_*)"""
    lo, hi = -2 ** 31, 2 ** 31 - 1
    rng = np.random.default_rng(31)
    pool = np.array([hi, lo, lo, hi, hi, lo + 1, hi - 1, 0, -1, 1, lo, lo, hi, hi, 2 ** 30, -2 ** 30, hi, lo, 5, lo, hi, -7], np.int64)
    rows = [pool, pool[::-1], np.resize(np.array([hi, lo]), pool.size), np.resize(np.array([lo, lo, hi]), pool.size)]
    rows += [rng.choice(pool, size=pool.size) for _ in range(60)]
    avg = np.array(rows, np.int64).astype(np.int32)
    for res in (np.zeros_like(avg), rng.choice(np.array([hi, lo, 0, 1, -1, 12345], np.int64), size=avg.shape).astype(np.int32)):
        assert_bits_equal(host.ModularChannel.inverseHorizontalSqueeze(ctx, avg, res), orc.inv_hsqueeze(avg, res), "overflow h")
        at, rt = avg.T.copy(), res.T.copy()
        assert_bits_equal(host.ModularChannel.inverseVerticalSqueeze(ctx, at, rt), orc.inv_vsqueeze(at, rt), "overflow v")
    # long rows: the segmented walk (warm-up from a guessed state, verify, redo) over the same extremes
    avg = rng.choice(pool, size=(70, 700)).astype(np.int32)
    res = rng.choice(np.array([hi, lo, 0, 3, -3], np.int64), size=(70, 700)).astype(np.int32)
    assert_bits_equal(host.ModularChannel.inverseHorizontalSqueeze(ctx, avg, res), orc.inv_hsqueeze(avg, res), "overflow h long")
    assert_bits_equal(host.ModularChannel.inverseVerticalSqueeze(ctx, avg.T.copy(), res.T.copy()),
                      orc.inv_vsqueeze(avg.T.copy(), res.T.copy()), "overflow v long")


def test_squeeze_shape_errors(ctx):
    with pytest.raises((ValueError, _lib.IllegalArgumentException)):
        host.ModularChannel.inverseHorizontalSqueeze(ctx, np.zeros((4, 5), np.int32), np.zeros((4, 3), np.int32))
    with pytest.raises((RuntimeError, _lib.IllegalStateException)):
        host.ModularChannel.inverseVerticalSqueeze(ctx, np.zeros((5, 4), np.int32), np.zeros((3, 4), np.int32))


@pytest.mark.parametrize("rct_type", [0, 1, 2, 3, 4, 5, 6, 7 + 3, 14 + 6, 35 + 5, 41])
def test_rct(ctx, orc, rct_type):
    v = np.random.default_rng(rct_type).integers(-2 ** 31, 2 ** 31 - 1, size=(3, 19, 23)).astype(np.int32)
    assert_bits_equal(host.rct(ctx, v, rct_type), orc.rct(v, rct_type), "rct %d" % rct_type)


def test_modular_to_float(ctx, orc):
    rng = np.random.default_rng(3)
    a = rng.integers(-70000, 70000, size=(31, 17)).astype(np.int32)
    b = rng.integers(-70000, 70000, size=(31, 17)).astype(np.int32)
    assert_bits_equal(host.modularToFloat(ctx, a, None, 0.0037), orc.modular_to_float(a, None, 0.0037), "to float")
    assert_bits_equal(host.modularToFloat(ctx, a, b, 1.0 / 255), orc.modular_to_float(a, b, 1.0 / 255), "to float sum")


@pytest.mark.parametrize("w,h,ch", [(1, 1, 3), (8, 8, 3), (9, 9, 1), (53, 37, 3), (37, 130, 4), (640, 360, 3), (1920, 1080, 3)])
def test_apply_transforms_default_plan(ctx, orc, w, h, ch, h_kernel):
    mod = synth.make_modular_frame(w, h, channels=ch, seed=w + h)
    ms = host.ModularStream(ctx, mod["chans"], mod["sp"])
    out = ms.applyTransforms()
    exp = orc.modular_apply(mod["chans"], mod["sp"])
    assert len(out) == len(exp) == ch
    for i, (a, b) in enumerate(zip(out, exp)):
        assert_bits_equal(a, b, "channel %d of %dx%d" % (i, w, h))
    assert ms.applyTransforms() is ms.channels  # second call is a no-op, like ModularStream.transformed


def test_apply_transforms_with_rct_and_rerun(ctx, orc):
    mod = synth.make_modular_frame(100, 60, channels=3, seed=5)
    ms = host.ModularStream(ctx, mod["chans"], mod["sp"], rctType=6 + 7 * 2, rctBegin=0)
    ms.begin()
    ms.run()
    first = ms.getDecodedBuffer()
    ms.run()  # re-runnable: inputs are not consumed
    second = ms.getDecodedBuffer()
    exp = orc.modular_apply(mod["chans"], mod["sp"], rct_type=6 + 7 * 2, rct_begin=0)
    for a, b, c in zip(first, second, exp):
        assert_bits_equal(a, c, "rct run 1")
        assert_bits_equal(b, c, "rct run 2")
    # RCT only (no squeeze): must not modify the caller's input
    v = [np.random.default_rng(i).integers(0, 256, size=(12, 20)).astype(np.int32) for i in range(3)]
    keep = [a.copy() for a in v]
    ms2 = host.ModularStream(ctx, v, [], rctType=10, rctBegin=0)
    out = ms2.applyTransforms()
    e = orc.rct(np.stack(keep), 10)
    for i in range(3):
        assert_bits_equal(out[i], e[i], "rct only")
        assert np.array_equal(v[i], keep[i])


def test_squeeze_round_trip_property_full_size(ctx, orc):
    """size-independent property at BASELINE's 8K Modular size: inverse(forward(x)) == x on the GPU
    (forward steps are the oracle's test-only forward squeeze)"""
    h, w, ch = 4320, 7680, 1
    rng = np.random.default_rng(7)
    img = rng.integers(0, 65536, size=(h, w)).astype(np.int32)
    sp = synth.default_squeeze_params([(h, w)] * ch)
    chans = [img]
    for (horiz, in_place, begin, num) in sp:
        end = begin + num - 1
        offset = end + 1 if in_place else len(chans)
        for k in range(begin, end + 1):
            a, r = (orc.fwd_hsqueeze if horiz else orc.fwd_vsqueeze)(chans[k])
            chans[k] = a
            chans.insert(offset + k - begin, r)
    out = host.ModularStream(ctx, chans, sp).applyTransforms()
    assert len(out) == 1 and np.array_equal(out[0], img)


# ---- the segmented walk (jxl_internal.h, kSqueezeSeg / kSqueezeWarm): one-step plans through jxl_modular_begin
def _one_step(ctx, orc, avg, res, horizontal):
    sp = [(1 if horizontal else 0, 1, 0, 1)]
    out = host.ModularStream(ctx, [avg, res], sp).applyTransforms()
    exp = orc.modular_apply([avg, res], sp)
    assert len(out) == len(exp) == 1
    return out[0], exp[0]


@pytest.fixture(params=["walk", "lds"])
def h_kernel(request, monkeypatch):
    """both forms of the H step: the register walk and the LDS-staged kernel (the library picks by step size)"""
    monkeypatch.setenv("JXL_HSQUEEZE_WALK_MAX", "0" if request.param == "lds" else str(1 << 40))
    return request.param


@pytest.mark.parametrize("n,other", [(65, 3), (128, 64), (129, 70), (500, 130), (1000, 5)])
@pytest.mark.parametrize("horizontal", [True, False])
def test_segmented_squeeze_random(ctx, orc, n, other, horizontal, h_kernel):
    """axis longer than one segment, odd and even totals, ragged last segment, several row / column blocks"""
    rng = np.random.default_rng(n * 7 + other)
    for odd in (0, 1):
        a = rng.integers(-3000, 3000, size=(other, n + odd)).astype(np.int32)
        r = np.rint(rng.laplace(0, 40, size=(other, n))).astype(np.int32)
        if not horizontal:
            a, r = a.T.copy(), r.T.copy()
        got, exp = _one_step(ctx, orc, a, r, horizontal)
        assert_bits_equal(got, exp, "segmented %s n=%d odd=%d" % ("h" if horizontal else "v", n, odd))


def _adversarial(n, other):
    """Rows whose recurrence never forgets its start: with avg falling by 1000 per pair and residual 1992 the chain sits
    in the slope-2 clamp of tendency(), where (left - avg) = u maps to 3 - u: the true walk cycles 2,1,2,1 (residual
    1994 on the first pair puts it there), a walk started from the guessed state cycles 0,3,0,3. Every segment boundary
    mismatches, so the verification kernel has to redo the rows serially. Row 1 is ordinary data."""
    a = np.empty((other, n), np.int64)
    a[:] = 10_000_000 - 1000 * np.arange(n)
    r = np.full((other, n), 1992, np.int64)
    r[:, 0] = 1994
    rng = np.random.default_rng(5)
    a[1] = rng.integers(-3000, 3000, size=n)
    r[1] = np.rint(rng.laplace(0, 40, size=n))
    return a.astype(np.int32), r.astype(np.int32)


@pytest.mark.parametrize("horizontal", [True, False])
def test_segmented_squeeze_adversarial(ctx, orc, horizontal, h_kernel):
    a, r = _adversarial(300, 70)
    # the construction does what it says: walking row 0 from the guess at pair 48 never meets the true walk
    t = orc.inv_hsqueeze(a[:1], r[:1])[0]
    u_true = (t[1::2][:-1].astype(np.int64) - a[0, 1:]) % 4
    assert set(u_true[40:80].tolist()) <= {1, 2}
    if not horizontal:
        a, r = a.T.copy(), r.T.copy()
    got, exp = _one_step(ctx, orc, a, r, horizontal)
    assert_bits_equal(got, exp, "adversarial %s" % ("h" if horizontal else "v"))


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["1", None])
@pytest.mark.parametrize("horizontal", [True, False])
def test_plan_with_speculative_checks_redoes_adversarial_rows(ctx, orc, horizontal, mode, monkeypatch):
    """jxl_modular_run checks the segmented walks without holding the next step back -- inside the next step's walk launch
    (the default, r4) or on a side stream (JXL_SQUEEZE_SPECULATE=1) -- and only reports a mismatch; reading the result then runs
    the plan again in order. Rows that never forget their start (see _adversarial) force that path: the output is still the
    serial walk's, bit for bit, and the redo is counted. Ordinary data right after it: no redo."""
    a, r = _adversarial(300, 70)
    exp = orc.inv_hsqueeze(a, r)
    if not horizontal:
        a, r, exp = a.T.copy(), r.T.copy(), exp.T.copy()
    sp = [(1 if horizontal else 0, 1, 0, 1)]
    if mode is None:
        monkeypatch.delenv("JXL_SQUEEZE_SPECULATE", raising=False)
    else:
        monkeypatch.setenv("JXL_SQUEEZE_SPECULATE", mode)
    before = ctx.lib.jxl_modular_redo_count(ctx.h)
    ms = host.ModularStream(ctx, [a, r], sp)
    out = ms.applyTransforms()
    assert len(out) == 1
    assert_bits_equal(out[0], exp, "adversarial plan %s" % ("h" if horizontal else "v"))
    assert ctx.lib.jxl_modular_redo_count(ctx.h) == before + 1
    rng = np.random.default_rng(9)
    a2 = rng.integers(-3000, 3000, size=a.shape).astype(np.int32)
    r2 = np.rint(rng.laplace(0, 40, size=r.shape)).astype(np.int32)
    exp2 = orc.inv_hsqueeze(a2, r2) if horizontal else orc.inv_hsqueeze(a2.T.copy(), r2.T.copy()).T
    out2 = host.ModularStream(ctx, [a2, r2], sp).applyTransforms()
    assert_bits_equal(out2[0], exp2, "ordinary plan")
    assert ctx.lib.jxl_modular_redo_count(ctx.h) == before + 1


@pytest.mark.gpu
@pytest.mark.parametrize("horizontal", [True, False])
def test_plan_in_order_repairs_adversarial_rows(ctx, orc, horizontal, monkeypatch):
    """JXL_SQUEEZE_SPECULATE=0 (the form a reported mismatch falls back to): the verification launch of a step repairs mismatching
    rows before the next step starts -- no second run"""
    monkeypatch.setenv("JXL_SQUEEZE_SPECULATE", "0")
    a, r = _adversarial(300, 70)
    exp = orc.inv_hsqueeze(a, r)
    if not horizontal:
        a, r, exp = a.T.copy(), r.T.copy(), exp.T.copy()
    before = ctx.lib.jxl_modular_redo_count(ctx.h)
    out = host.ModularStream(ctx, [a, r], [(1 if horizontal else 0, 1, 0, 1)]).applyTransforms()
    assert_bits_equal(out[0], exp, "adversarial plan %s" % ("h" if horizontal else "v"))
    assert ctx.lib.jxl_modular_redo_count(ctx.h) == before


# ---- r5: a V step and the H step behind it as one launch (k_modular_vh.hip) ---------------------------------------------------
def _vh_inputs(rng, htot, wtot, nch=1, big=False):
    """channel list of a two-step plan (forward order H then V, so the inverse runs V then H): avg, V residuals, H residuals"""
    ah, rh, aw, rw = (htot + 1) // 2, htot // 2, (wtot + 1) // 2, wtot // 2
    lo, hi = (-2 ** 31, 2 ** 31 - 1) if big else (-3000, 3000)
    avg = [rng.integers(lo, hi, size=(ah, aw)).astype(np.int32) for _ in range(nch)]
    if big:
        vr = [rng.integers(lo, hi, size=(rh, aw)).astype(np.int32) for _ in range(nch)]
        hr = [rng.integers(lo, hi, size=(htot, rw)).astype(np.int32) for _ in range(nch)]
    else:
        vr = [np.rint(rng.laplace(0, 40, size=(rh, aw))).astype(np.int32) for _ in range(nch)]
        hr = [np.rint(rng.laplace(0, 40, size=(htot, rw))).astype(np.int32) for _ in range(nch)]
    return avg + vr + hr, [(1, 1, 0, nch), (0, 1, 0, nch)]


@pytest.mark.parametrize("htot,wtot", [(2, 2), (3, 3), (5, 34), (64, 32), (64, 66), (65, 65), (66, 130), (127, 97), (128, 256), (130, 257),
                                       (193, 67), (200, 1030), (321, 514), (1080, 1920)])
@pytest.mark.parametrize("seg,cw", [(None, None), ("16", "16"), ("32", "16"), ("64", "16"), ("32", "32"), ("64", "32"), (None, "32")])
def test_fused_vh_pair(ctx, orc, htot, wtot, seg, cw, monkeypatch):
    """V + H of one level in one launch: every edge the tile walk has -- one stripe / several, ragged last stripe, odd height (the
    copied last row) and odd width (the copied last column), one segment / several with a ragged last one, the extra chunk that
    only holds the odd column -- against the oracle's two serial steps"""
    for name, val in (("JXL_VH_SEG", seg), ("JXL_VH_CW", cw)):  # cw: chunk width of the kernel (16 / 32 H pairs), None = the library's choice
        if val is None:
            monkeypatch.delenv(name, raising=False)
        else:
            monkeypatch.setenv(name, val)
    rng = np.random.default_rng(htot * 131 + wtot)
    chans, sp = _vh_inputs(rng, htot, wtot)
    ms = host.ModularStream(ctx, chans, sp)
    out = ms.applyTransforms()
    exp = orc.modular_apply(chans, sp)
    assert len(out) == len(exp) == 1 and out[0].shape == (htot, wtot)
    assert_bits_equal(out[0], exp[0], "fused V+H %dx%d seg %s cw %s" % (htot, wtot, seg, cw))
    assert ctx.lib.jxl_modular_last_launch_count(ctx.h) <= 2  # the pair + (at most) one check launch: the fused kernel ran


@pytest.mark.parametrize("cw", ["16", "32"])
def test_fused_vh_three_channels_and_rerun(ctx, orc, cw, monkeypatch):
    monkeypatch.setenv("JXL_VH_SEG", "32")
    monkeypatch.setenv("JXL_VH_CW", cw)
    rng = np.random.default_rng(77)
    chans, sp = _vh_inputs(rng, 150, 201, nch=3)
    ms = host.ModularStream(ctx, chans, sp)
    ms.begin()
    exp = orc.modular_apply(chans, sp)
    for run in range(2):
        ms.run()
        got = ms.getDecodedBuffer()
        assert len(got) == 3
        for i in range(3):
            assert_bits_equal(got[i], exp[i], "channel %d run %d" % (i, run))


def test_fused_vh_int32_extremes_take_the_exact_path(ctx, orc):
    """operands outside the range guard of the short tendency form (modular_tend.h): the chunk is walked with the reference's
    long form; wrap-around as in Java"""
    rng = np.random.default_rng(3)
    chans, sp = _vh_inputs(rng, 130, 140, big=True)
    before = ctx.lib.jxl_modular_redo_count(ctx.h)
    out = host.ModularStream(ctx, chans, sp).applyTransforms()
    assert_bits_equal(out[0], orc.modular_apply(chans, sp)[0], "extremes")
    # mixed: ordinary data with a few extreme samples
    chans, sp = _vh_inputs(rng, 130, 140)
    chans[0][5, 7] = 2 ** 31 - 1
    chans[1][9, 3] = -2 ** 31
    chans[2][100, 50] = 2 ** 30
    out = host.ModularStream(ctx, chans, sp).applyTransforms()
    assert_bits_equal(out[0], orc.modular_apply(chans, sp)[0], "mixed extremes")
    assert ctx.lib.jxl_modular_redo_count(ctx.h) >= before  # (white-noise extremes may or may not forget their start)


@pytest.mark.parametrize("cw", ["16", "32"])
@pytest.mark.parametrize("axis", ["h", "v"])
def test_fused_vh_reports_adversarial_chains_and_redoes_in_order(ctx, orc, axis, cw, monkeypatch):
    """chains that never forget their start (see _adversarial) across an H segment boundary / a V stripe or quarter boundary: the
    fused launch reports, the plan runs again with the one-step kernels in order, and the result is the serial walk's"""
    monkeypatch.setenv("JXL_VH_SEG", "32")
    monkeypatch.setenv("JXL_VH_CW", cw)
    rng = np.random.default_rng(21)
    chans, sp = _vh_inputs(rng, 200, 300)
    if axis == "h":  # the V output is what the H step sees as averages: the forward V step of the adversarial rows
        a, r = _adversarial(150, 200)              # H step: 200 rows, 150 pairs
        vavg, vres = orc.fwd_vsqueeze(a)
        chans = [vavg, vres, r]
    else:
        a, r = _adversarial(100, 150)              # as columns: 100 V pairs, 150 columns
        chans = [a.T.copy(), r.T.copy(), np.rint(rng.laplace(0, 40, size=(200, 150))).astype(np.int32)]
    before = ctx.lib.jxl_modular_redo_count(ctx.h)
    out = host.ModularStream(ctx, chans, sp).applyTransforms()
    assert_bits_equal(out[0], orc.modular_apply(chans, sp)[0], "adversarial %s" % axis)
    assert ctx.lib.jxl_modular_redo_count(ctx.h) == before + 1


def test_fused_plan_equals_unfused_plan(ctx, orc, monkeypatch):
    """the default plan of a 3-channel image with and without the pair kernel (JXL_SQUEEZE_NO_VH): same samples, fewer launches"""
    mod = synth.make_modular_frame(611, 437, channels=3, seed=4)
    exp = orc.modular_apply(mod["chans"], mod["sp"])
    ms = host.ModularStream(ctx, mod["chans"], mod["sp"])
    out = ms.applyTransforms()
    fused_launches = ctx.lib.jxl_modular_last_launch_count(ctx.h)
    monkeypatch.setenv("JXL_SQUEEZE_NO_VH", "1")
    ms2 = host.ModularStream(ctx, mod["chans"], mod["sp"])
    out2 = ms2.applyTransforms()
    for i in range(3):
        assert_bits_equal(out[i], exp[i], "fused channel %d" % i)
        assert_bits_equal(out2[i], exp[i], "unfused channel %d" % i)
    assert fused_launches < ctx.lib.jxl_modular_last_launch_count(ctx.h)


def test_fused_plan_repeats_bit_for_bit(ctx, orc):
    """the same 1080p plan five times, every run against the oracle: the r5 store-data hazard (k_modular_vh.hip, vh_store: a VALU write
    to the data registers of a 16-byte buffer store that has just issued) corrupted a few dozen samples per image, different ones from
    run to run -- a parity test that runs a plan once can pass by luck"""
    mod = synth.make_modular_frame(1920, 1080, channels=3, seed=3000)
    exp = orc.modular_apply(mod["chans"], mod["sp"])
    ms = host.ModularStream(ctx, mod["chans"], mod["sp"])
    ms.begin()
    for run in range(5):
        ms.run()
        got = ms.getDecodedBuffer()
        for i in range(3):
            assert_bits_equal(got[i], exp[i], "run %d channel %d" % (run, i))

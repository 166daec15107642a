"""GPU parity of the VarDCT path: HIP kernels (through the C-ABI) vs the CPU oracle, bit-exact.

Bar (BASELINE.json north_star): VarDCT/XYB within 1 ulp float -- we hold the stronger bar, bit
identity with the oracle, for every stage up to and including invertXYB; only the PQ/sRGB transfer
stage (double pow) is allowed 1 ulp.
"""
import numpy as np
import pytest

from conftest import assert_bits_equal
from jxlatte_amd import _lib, abi, host, synth

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


# ---- row a16: chroma-subsampled (JPEG-recompression) frames -------------------------------------------------------------
SUBSAMPLINGS = {"420": ((1, 0, 1), (1, 0, 1)), "422": ((0, 0, 0), (1, 0, 1)), "440": ((1, 0, 1), (0, 0, 0)),
                "luma_sub": ((0, 1, 0), (0, 1, 0))}


@pytest.mark.gpu
@pytest.mark.parametrize("mode", sorted(SUBSAMPLINGS))
@pytest.mark.parametrize("size", [(64, 32), (272, 48), (528, 272)])
def test_chroma_subsampled_frame(ctx, orc, mode, size):
    """IDCT on each channel's own grid, chroma-from-luma skipped, Frame.invertSubsampling, then Gab / EPF on full planes"""
    sy, sx = SUBSAMPLINGS[mode]
    base = synth.make_vardct_frame(size[0], size[1], seed=size[0] + len(mode), mix="dct8", xyb=0)
    fr = synth.make_subsampled(base, sy, sx)
    for stages in (abi.STAGE_IDCT, abi.STAGE_IDCT | abi.STAGE_GAB | abi.STAGE_EPF):
        got = host.Frame.from_synth(ctx, fr, stages=stages).decodeFrame()
        exp = orc.vardct_frame(fr, stages=stages)
        assert_bits_equal(got, exp, "%s %s stages %d" % (mode, size, stages))
    # the surviving chroma really differs from the unsubsampled decode (the test would be vacuous otherwise)
    plain = orc.vardct_frame(base, stages=abi.STAGE_IDCT)
    assert not np.array_equal(plain[0 if mode != "luma_sub" else 1], exp[0 if mode != "luma_sub" else 1])


@pytest.mark.gpu
def test_chroma_subsampled_lfquant_and_errors(ctx, orc):
    from jxlatte_amd import _lib
    base = synth.make_vardct_frame(64, 32, seed=5, mix="dct8", xyb=0)
    fr = synth.make_subsampled(base, (1, 0, 1), (1, 0, 1))
    f = host.Frame.from_synth(ctx, fr, stages=abi.STAGE_IDCT)
    q = [np.ascontiguousarray(np.rint(g * 64).astype(np.int32)) for g in fr["lfgroups"][0]["lf"]]
    with pytest.raises(_lib.InvalidBitstreamException):  # LFCoefficients.java:36-37
        d = abi.LFQuantDesc()
        d.cells_h, d.cells_w = 4, 8
        for c in range(3):
            d.lf_quant[c] = abi.iptr(q[c])
            d.scaled_dequant[c] = 1.0 / 64
        d.adaptive_smoothing = 1
        ctx.call("jxl_vardct_set_lfgroup_lfquant", __import__("ctypes").byref(d))
    d.adaptive_smoothing = 0
    ctx.call("jxl_vardct_set_lfgroup_lfquant", __import__("ctypes").byref(d))
    got = f.decodeFrame()
    fr2 = dict(fr)
    g2 = dict(fr["lfgroups"][0])
    g2["lf"] = [np.ascontiguousarray(a.astype(np.float32) * np.float32(1.0 / 64)) for a in q]
    fr2["lfgroups"] = [g2]
    assert_bits_equal(got, orc.vardct_frame(fr2, stages=abi.STAGE_IDCT), "subsampled LF quant")


def test_run_batch_equals_single_runs(orc):
    """jxl_vardct_run_batch: frames of different sizes, mixes and restoration settings in one batch, twice (argument
    blocks cached on the second call), then with one frame replaced (cache rebuilt); every frame equals the oracle"""
    specs = [dict(w=256, h=128, seed=21, mix="all"), dict(w=512, h=256, seed=22, mix="default"), dict(w=136, h=72, seed=23, mix="dct8"),
             dict(w=320, h=192, seed=24, mix="default", epf_iters=1)]
    ctxs = [_lib.Context(0) for _ in specs]
    try:
        synths = [synth.make_vardct_frame(sp["w"], sp["h"], seed=sp["seed"], mix=sp["mix"], epf_iters=sp.get("epf_iters", 2)) for sp in specs]
        frames = [host.Frame.from_synth(c, f) for c, f in zip(ctxs, synths)]
        exp = [orc.vardct_frame(f) for f in synths]
        for _ in range(2):
            host.Frame.runBatch(frames)
            for i, fr in enumerate(frames):
                assert_bits_equal(fr.readOutput(), exp[i], "batch frame %d" % i)
        # a new frame in context 1 (same context, new tables): the cached argument blocks must not be reused
        synths[1] = synth.make_vardct_frame(384, 128, seed=29, mix="all")
        frames[1] = host.Frame.from_synth(ctxs[1], synths[1])
        exp[1] = orc.vardct_frame(synths[1])
        host.Frame.runBatch(frames)
        for i, fr in enumerate(frames):
            assert_bits_equal(fr.readOutput(), exp[i], "batch frame %d after replacement" % i)
        # frames 0..2 share one restoration variant (Gab + EPF x2, float out): the restoration stage is one launch too
        for _ in range(2):
            host.Frame.runBatch(frames[:3])
            for i, fr in enumerate(frames[:3]):
                assert_bits_equal(fr.readOutput(), exp[i], "same-variant batch frame %d" % i)
        # a batch holding a frame the shared launches do not cover (128-edge blocks) falls back to single runs
        synths[2] = synth.make_vardct_frame(512, 256, seed=31, mix="large")
        frames[2] = host.Frame.from_synth(ctxs[2], synths[2])
        exp[2] = orc.vardct_frame(synths[2])
        host.Frame.runBatch(frames)
        for i, fr in enumerate(frames):
            assert_bits_equal(fr.readOutput(), exp[i], "fallback batch frame %d" % i)
    finally:
        for c in ctxs:
            c.close()


@pytest.mark.parametrize("mix,aligned", [("all", True), ("default", True), ("all", False), ("large", True)])
@pytest.mark.parametrize("nonzero_p", [0.0, 0.004, 0.03])
def test_sparse_coefficients_and_signed_zero_lf(ctx, orc, mix, aligned, nonzero_p):
    """Zero skipping in the IDCT MAC loops (MirrorAcc::step_sparse): frames whose coefficient rows / columns are mostly
    zero, with LF samples that are +0.0, -0.0 (the one start value for which adding a skipped +0 product would matter)
    and ordinary values mixed, IDCT stage alone and the whole path; bit-identical to the oracle, signs of zeros included"""
    W, H = (1024, 512) if mix == "large" else (384, 256)
    frame = synth.make_vardct_frame(W, H, seed=77 + int(nonzero_p * 1000), mix=mix, aligned=aligned, nonzero_p=nonzero_p)
    rng = np.random.default_rng(5)
    for g in frame["lfgroups"]:
        for c in range(3):
            lf = g["lf"][c]
            r = rng.random(lf.shape)
            lf[r < 0.25] = -0.0
            lf[(r >= 0.25) & (r < 0.45)] = 0.0
    for stages in (abi.STAGE_IDCT, None):
        fr = host.Frame.from_synth(ctx, frame, stages=stages)
        got = fr.decodeFrame()
        exp = orc.vardct_frame(frame, stages=stages)
        assert_bits_equal(got, exp, "sparse %s p=%g stages=%s" % (mix, nonzero_p, stages))


# ---- the PCIe leg: int16 wire format and page-locked buffers ---------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("pinned", [False, True, "misaligned"])
def test_int16_wire_format_and_pinned_buffers(ctx, orc, pinned):
    """jxl_vardct_put_group_i16 (+ page-locked sources from jxl_host_alloc, which the device reads in place when they are
    16-byte aligned -- "misaligned": a page-locked source that is not, which goes through the staging ring like pageable
    memory; 12 puts wrap the ring of 8) fills the same coefficient planes as jxl_vardct_put_group: identical frame output,
    also for a second pass that accumulates (PassGroup.java:174-200)"""
    from jxlatte_amd import _lib
    fr = synth.make_vardct_frame(520, 264, seed=77, aligned=False)
    assert np.abs(fr["coeff"]).max() < 32768
    p = abi.VarDCTParams.from_buffer_copy(fr["params"])
    p.stages = abi.STAGE_IDCT | abi.STAGE_GAB | abi.STAGE_EPF
    lib = _lib.load()
    keep = []

    def feed(use16, passes):
        f = host.Frame(ctx, p, fr["weights"], fr["woffs"])
        for g in fr["lfgroups"]:
            f.setLFGroup(g)
        for grp in range(synth.num_groups(fr)):
            planes = synth.group_view(fr, grp)
            for ps in range(passes):
                # pass 0 carries q - 3 * (passes - 1), every later pass adds 3: the sum is q
                part = [(a - 3 * (passes - 1) if ps == 0 else np.full_like(a, 3)) for a in planes]
                if pinned:
                    dt = np.int16 if use16 else np.int32
                    pad = 1 if pinned == "misaligned" else 0
                    pa = [host.PinnedArray(lib, (a.shape[0], a.shape[1] + pad), dt) for a in part]
                    for dst, a in zip(pa, part):
                        dst.array[:, pad:] = a
                    keep.extend(pa)
                    part = [x.array[:, pad:] for x in pa]
                if use16:
                    f.putGroupI16(ps, grp, part)
                else:
                    f.putGroup(ps, grp, part)
        return f.decodeFrame()

    ref = feed(False, 1)
    assert_bits_equal(ref, orc.vardct_frame(fr, stages=p.stages), "int32 path")
    assert_bits_equal(feed(True, 1), ref, "int16 wire format")
    assert_bits_equal(feed(True, 2), ref, "int16, two passes")
    assert_bits_equal(feed(False, 2), ref, "int32, two passes")
    for x in keep:
        x.free()


@pytest.mark.gpu
def test_mapped_int16_coefficient_planes(ctx, orc):
    """jxl_vardct_map_coeffs_i16 / commit: groups written in place into the library's page-locked planes give the same frame;
    a put_group after the commit overrides its rectangle (the int32 fallback for a group that does not fit 16 bits)"""
    fr = synth.make_vardct_frame(520, 264, seed=78, aligned=False)
    p = abi.VarDCTParams.from_buffer_copy(fr["params"])
    p.stages = abi.STAGE_IDCT | abi.STAGE_GAB | abi.STAGE_EPF
    big = dict(fr)
    big["coeff"] = fr["coeff"].copy()
    big["coeff"][1, 3, 5] = 70000  # one sample of group 0 outside the int16 range
    exp = orc.vardct_frame(big, stages=p.stages)
    f = host.Frame(ctx, p, fr["weights"], fr["woffs"])
    for g in fr["lfgroups"]:
        f.setLFGroup(g)
    planes = f.mapCoeffsI16()
    assert all(int(np.abs(a).max()) == 0 for a in planes)  # zero-filled at map time
    for c in range(3):
        planes[c][...] = np.clip(big["coeff"][c], -32768, 32767)
    f.commitCoeffsI16()
    f.putGroup(0, 0, synth.group_view(big, 0))  # group 0 again, 32-bit
    assert_bits_equal(f.decodeFrame(), exp, "mapped int16 planes + int32 fallback group")
    with pytest.raises(_lib.JxlError):
        f2 = host.Frame(ctx, p, fr["weights"], fr["woffs"])
        f2.commitCoeffsI16()  # nothing mapped for this frame


@pytest.mark.gpu
def test_mapped_planes_without_zero_fill_and_split_read_output(ctx, orc):
    """jxl_vardct_map_coeffs_i16_ex(JXL_MAP_NO_FILL) + commit_..._groups: the planes come back dirty, the groups the caller names
    count, every other group reads as zero (the reference's fresh int[][], HFCoefficients.java:68); jxl_vardct_read_output_begin /
    _wait return what read_output returns, with the next frame of the same context driven in between"""
    fr = synth.make_vardct_frame(520, 520, seed=79, aligned=False)  # 3 x 3 groups
    p = abi.VarDCTParams.from_buffer_copy(fr["params"])
    p.stages = abi.STAGE_IDCT | abi.STAGE_GAB | abi.STAGE_EPF
    n_groups = synth.num_groups(fr)
    assert n_groups == 9
    written = np.ones(n_groups, np.uint8)
    written[[2, 4]] = 0
    # expected: the coefficients of the unwritten groups are zero
    zeroed = dict(fr)
    zeroed["coeff"] = fr["coeff"].copy()
    for g in (2, 4):
        gy, gx = divmod(g, 3)
        zeroed["coeff"][:, gy * 256:(gy + 1) * 256, gx * 256:(gx + 1) * 256] = 0
    exp = orc.vardct_frame(zeroed, stages=p.stages)
    exp_full = orc.vardct_frame(fr, stages=p.stages)
    f = host.Frame(ctx, p, fr["weights"], fr["woffs"])
    for g in fr["lfgroups"]:
        f.setLFGroup(g)
    # dirty the staging buffer first: a zero-filling map, garbage stores, no commit
    planes = f.mapCoeffsI16()
    for c in range(3):
        planes[c][...] = 12345
    planes = f.mapCoeffsI16(no_fill=True)
    assert all(int(a.min()) == 12345 for a in planes)  # not zero-filled
    for c in range(3):
        planes[c][...] = fr["coeff"][c]  # every group written, two of them not named below
    with pytest.raises(_lib.JxlError):
        f.commitCoeffsI16()  # planes mapped without fill need the group list
    with pytest.raises(_lib.JxlError):
        f.commitCoeffsI16(written[:-1])  # wrong number of groups
    f.commitCoeffsI16(written)
    f.run()
    got = f.readOutputBegin()
    # the next frame of the same context while the copy is in flight: all groups written
    f2 = host.Frame(ctx, p, fr["weights"], fr["woffs"])
    for g in fr["lfgroups"]:
        f2.setLFGroup(g)
    planes = f2.mapCoeffsI16(no_fill=True)
    for c in range(3):
        planes[c][...] = fr["coeff"][c]
    f2.commitCoeffsI16(np.ones(n_groups, np.uint8))
    f.readOutputWait()
    assert_bits_equal(got, exp, "unwritten groups read as zero")
    assert_bits_equal(f2.decodeFrame(), exp_full, "all groups written, no zero-fill")
    with pytest.raises(_lib.JxlError):
        f2.readOutputWait()  # nothing begun


@pytest.mark.gpu
def test_groups_never_put_read_as_zero_also_on_a_reused_context(ctx, orc):
    """begin_frame gives a frame fresh coefficient planes (HFCoefficients.java:68: new int[..]); since r4 the zero-fill is deferred
    until something needs it. A frame whose caller puts only some groups must see zeros in the others -- also when the context
    has just run another frame whose coefficients are still in the planes, through the per-group entries (int32 and int16) and
    when no group is put at all"""
    fr = synth.make_vardct_frame(520, 520, seed=80, aligned=False)  # 3 x 3 groups
    p = abi.VarDCTParams.from_buffer_copy(fr["params"])
    p.stages = abi.STAGE_IDCT | abi.STAGE_GAB | abi.STAGE_EPF
    full = orc.vardct_frame(fr, stages=p.stages)

    def frame(put, i16=False):
        f = host.Frame(ctx, p, fr["weights"], fr["woffs"])
        for g in fr["lfgroups"]:
            f.setLFGroup(g)
        for grp in put:
            q = synth.group_view(fr, grp)
            if i16:
                f.putGroupI16(0, grp, [np.ascontiguousarray(a, np.int16) for a in q])
            else:
                f.putGroup(0, grp, q)
        return f.decodeFrame()

    def expected(put):
        z = dict(fr)
        z["coeff"] = np.zeros_like(fr["coeff"])
        for grp in put:
            gy, gx = divmod(grp, 3)
            z["coeff"][:, gy * 256:(gy + 1) * 256, gx * 256:(gx + 1) * 256] = fr["coeff"][:, gy * 256:(gy + 1) * 256, gx * 256:(gx + 1) * 256]
        return orc.vardct_frame(z, stages=p.stages)

    assert_bits_equal(frame(range(9)), full, "every group put")
    assert_bits_equal(frame([0, 4, 8]), expected([0, 4, 8]), "three groups put behind a full frame on the same context")
    assert_bits_equal(frame([]), expected([]), "no group put at all")
    assert_bits_equal(frame(range(9), i16=True), full, "every group put (int16)")
    assert_bits_equal(frame([1, 5], i16=True), expected([1, 5]), "two groups put (int16) behind a full frame")
    # mapped planes committed, then a later frame with per-group puts only
    f = host.Frame(ctx, p, fr["weights"], fr["woffs"])
    for g in fr["lfgroups"]:
        f.setLFGroup(g)
    planes = f.mapCoeffsI16(no_fill=True)
    for c in range(3):
        planes[c][...] = fr["coeff"][c]
    f.commitCoeffsI16(np.ones(9, np.uint8))
    assert_bits_equal(f.decodeFrame(), full, "mapped planes")
    assert_bits_equal(frame([2]), expected([2]), "one group put behind a frame committed from mapped planes")


@pytest.mark.gpu
def test_frame_geometry_queries(ctx):
    """jxl_vardct_geometry / jxl_vardct_group_size (r5): the numbers the JNI shim sizes its buffer checks from -- plane sizes and
    shifts per channel (paddedSize >> jpegUpsampling, HFCoefficients.java:64-69), the output sample size, and the rectangle of a
    group (Frame.getGroupLocation / getGroupSize, J/frame/Frame.java:767-786) incl. the ragged last column and row of groups"""
    import ctypes as C
    base = synth.make_vardct_frame(528, 272, seed=5, mix="dct8", xyb=0)
    fr = synth.make_subsampled(base, (1, 0, 1), (1, 0, 0))
    f = host.Frame.from_synth(ctx, fr, stages=abi.STAGE_IDCT)
    info = (C.c_int32 * 13)()
    ctx.call("jxl_vardct_geometry", info)
    W, H = f.width, f.height
    assert list(info)[:6] == [W >> 1, W, W, H >> 1, H, H >> 1]
    assert list(info)[6:12] == [1, 1, 0, 0, 0, 1] and info[12] == 4
    gw, gh = (C.c_int32 * 3)(), (C.c_int32 * 3)()
    grs, gcs = (W + 255) // 256, (H + 255) // 256
    ctx.call("jxl_vardct_group_size", 0, gw, gh)
    assert list(gw) == [128, 256, 256] and list(gh) == [128, 256, 128]
    ctx.call("jxl_vardct_group_size", grs * gcs - 1, gw, gh)
    lw, lh = W - 256 * (grs - 1), H - 256 * (gcs - 1)
    assert list(gw) == [lw >> 1, lw, lw] and list(gh) == [lh >> 1, lh, lh >> 1]
    from jxlatte_amd import _lib
    with pytest.raises(_lib.IllegalArgumentException):
        ctx.call("jxl_vardct_group_size", grs * gcs, gw, gh)

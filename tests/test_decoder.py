"""Row f2: the C++ bitstream front-end and the JXLDecoder host layer on REAL .jxl files (tests/golden/samples/, copies of
the reference's samples/ data files).

CPU: every sample parses with valid ANS final states; decoded pixels through the oracle-backed backend are pinned by
CRCs (regression) and by structural properties (white.jxl is white, art.jxl is the 128x128 8-bit RGB image of config C1).
GPU: the same files through the device library give bit-identical pixels to the oracle-backed decode -- parity of the
hot path on real varblock statistics, real LF images and real modular streams."""
import io
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import assert_bits_equal
from jxlatte_amd import frontend
from jxlatte_amd.decoder import JXLDecoder, PNGWriter, UnsupportedOperationException

SAMPLES = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "samples")
ALL = ["art", "quilt", "white", "blendmodes_5", "wb-rainbow", "lenna", "bbb", "patches-lossless", "bench"]
DECODABLE = ["art", "quilt", "white", "blendmodes_5", "wb-rainbow", "lenna", "bbb", "patches-lossless", "bench"]


def path(name):
    return os.path.join(SAMPLES, name + ".jxl")


@pytest.fixture(scope="module")
def oracle_backend(orc):
    from oracle.pybackend import OracleBackend
    return OracleBackend()


def oracle_hooks(orc):
    return (lambda ins, steps, shapes: orc.modular_apply(ins, steps, rct_type=-1, out_shapes=shapes),
            lambda a, b, c, t: orc.rct(np.stack([a, b, c]), t))


def read_png(data):
    pos, idat, hdr = 8, b"", None
    while pos < len(data):
        n = struct.unpack(">I", data[pos:pos + 4])[0]
        tag, body = data[pos + 4:pos + 8], data[pos + 8:pos + 8 + n]
        assert zlib.crc32(tag + body) & 0xffffffff == struct.unpack(">I", data[pos + 8 + n:pos + 12 + n])[0]
        if tag == b"IHDR":
            hdr = struct.unpack(">IIBBBBB", body)
        if tag == b"IDAT":
            idat += body
        pos += 12 + n
    w, h, bd, cm = hdr[:4]
    ch = {0: 1, 2: 3, 4: 2, 6: 4}[cm]
    raw = np.frombuffer(zlib.decompress(idat), np.uint8).reshape(h, 1 + w * ch * (bd // 8))
    assert not raw[:, 0].any()
    px = raw[:, 1:]
    return (px.reshape(h, w, ch) if bd == 8 else px.reshape(h, w, ch, 2).astype(np.uint16) @ np.array([256, 1], np.uint16)), bd


# ---- front-end only ------------------------------------------------------------------------------------------
EXPECT_HEADERS = {  # name: (width, height, bits, extra channels, xyb, orientation, frames, first frame encoding)
    "art": (128, 128, 8, 0, 0, 1, 1, 1), "quilt": (1024, 1024, 8, 0, 0, 1, 1, 1), "white": (320, 240, 8, 0, 1, 1, 1, 0),
    "blendmodes_5": (1024, 1024, 12, 1, 0, 1, 5, 1), "wb-rainbow": (2048, 1152, 9, 1, 0, 1, 5, 1),
    "lenna": (512, 512, 8, 0, 1, 1, 1, 0), "bbb": (1280, 720, 8, 0, 1, 1, 1, 0),
    "patches-lossless": (1600, 1096, 8, 1, 0, 1, 2, 1), "bench": (500, 606, 8, 0, 0, 5, 1, 0),
}


@pytest.mark.parametrize("name", ALL)
def test_frontend_parses_sample(orc, name):
    """every section of every frame decodes and every ANS stream ends in its initial state (the front-end checks
    EntropyStream.validateFinalState after each stream and fails otherwise)"""
    fe = frontend.Frontend(open(path(name), "rb").read())
    im = fe.image
    sq, rct = oracle_hooks(orc)
    frames = []
    while True:
        fr = fe.next_frame(sq, rct)
        if fr is None:
            break
        frames.append((fr.encoding, fr.width, fr.height, fr.num_groups))
    w, h, bits, extra, xyb, orient, nframes, enc0 = EXPECT_HEADERS[name]
    assert (im.width, im.height, im.bits_per_sample, im.num_extra, im.xyb_encoded, im.orientation) == (w, h, bits, extra, xyb, orient)
    assert len(frames) == nframes and frames[0][0] == enc0


def test_frontend_boundary_tensors_of_lenna(orc):
    fe = frontend.Frontend(open(path("lenna"), "rb").read())
    fr = fe.next_frame(None, None)  # a VarDCT frame needs no modular hooks
    assert (fr.width, fr.height, fr.num_groups, fr.num_lf_groups, fr.num_passes) == (512, 512, 4, 1, 1)
    g = fe.lfgroup(0)
    assert g["dct_select"].shape == (64, 64) and not (g["dct_select"] == 255).any()  # every cell is covered by a varblock
    # the block list reproduces the map: each listed corner carries its own type and the footprints tile the LF group
    from jxlatte_amd import abi
    cover = np.zeros((64, 64), np.int32)
    for (y, x) in g["block_yx"]:
        t = g["dct_select"][y, x]
        hh, ww = abi.TRANSFORM_TYPES[t][5] >> 3, abi.TRANSFORM_TYPES[t][6] >> 3
        assert (g["dct_select"][y:y + hh, x:x + ww] == t).all()
        cover[y:y + hh, x:x + ww] += 1
    assert (cover == 1).all()
    assert 0 <= g["sharpness"].min() and g["sharpness"].max() <= 7 and g["hf_mul"].min() >= 1
    q = fe.coeffs(0, 0)
    assert [a.shape for a in q] == [(256, 256)] * 3 and sum(int(np.count_nonzero(a)) for a in q) > 1000
    # LLF corners of the coefficient planes are never written by the HF decoder
    y, x = g["block_yx"][0]
    assert all(a[8 * y, 8 * x] == 0 for a in q)


def test_frontend_rejects_garbage():
    with pytest.raises(frontend.FrontendError):
        frontend.Frontend(b"\xff\x0b" + bytes(30))
    good = open(path("lenna"), "rb").read()
    fe = frontend.Frontend(good[:2000])  # header parses, the frame payload is truncated
    with pytest.raises(frontend.FrontendError):
        fe.next_frame(None, None)


def test_frontend_survives_mutated_input(orc):
    """corrupt bitstreams end in FrontendError (or decode to something), never in a crash; the sanitizer build of the same
    loop is `make -C jxlatte_amd/frontend fuzz`"""
    rng = np.random.default_rng(2026)
    sq, rct = oracle_hooks(orc)
    for name in ("art", "white", "blendmodes_5", "lenna"):
        good = bytearray(open(path(name), "rb").read())
        for _ in range(40):
            bad = bytearray(good)
            kind = rng.integers(0, 3)
            if kind == 0:
                for _k in range(int(rng.integers(1, 4))):
                    bad[int(rng.integers(0, min(len(bad), 300)))] ^= 1 << int(rng.integers(0, 8))
            elif kind == 1:
                bad = bad[:int(rng.integers(1, len(bad)))]
            else:
                pos = int(rng.integers(0, len(bad)))
                bad[pos:pos + 8] = bytes(rng.integers(0, 256, min(8, len(bad) - pos)).astype(np.uint8))
            try:
                fe = frontend.Frontend(bytes(bad))
                n = 0
                while fe.next_frame(sq, rct) is not None and n < 20:
                    n += 1
            except (frontend.FrontendError, RuntimeError, ValueError):
                pass


def test_frame_level_modular_needs_hooks():
    """no CPU fallback for the frame-level inverse Squeeze / RCT: without hooks the front-end refuses"""
    for name in ("art", "quilt"):  # art.jxl: frame-level RCT; quilt.jxl: frame-level Squeeze (16 steps)
        fe = frontend.Frontend(open(path(name), "rb").read())
        with pytest.raises(frontend.FrontendError, match="device hook"):
            fe.next_frame(None, None)


# ---- decoder through the oracle backend (CPU) --------------------------------------------------------------------
def decode(name, backend):
    dec = JXLDecoder(path(name), backend=backend)
    return dec, dec.decode()


def test_c1_art_jxl_to_png(oracle_backend):
    """config C1: samples/art.jxl -> a 128x128 8-bit RGB PNG"""
    _, img = decode("art", oracle_backend)
    assert (img.getWidth(), img.getHeight(), img.getColorChannelCount(), img.hasAlpha()) == (128, 128, 3, False)
    buf = io.BytesIO()
    PNGWriter(img).write(buf)
    px, bd = read_png(buf.getvalue())
    assert px.shape == (128, 128, 3) and bd == 8
    # lossless modular, sRGB tagged, 8 bit: the PNG samples are the decoded integers themselves
    for c in range(3):
        assert np.array_equal(px[..., c], np.clip(img.buffer[c], 0, 255))
    assert len(np.unique(px.reshape(-1, 3), axis=0)) > 50  # a picture, not a flat field


def test_white_jxl_is_white(oracle_backend):
    _, img = decode("white", oracle_backend)
    buf = io.BytesIO()
    PNGWriter(img).write(buf)
    px, _ = read_png(buf.getvalue())
    assert px.shape == (240, 320, 3) and px.min() >= 254


def test_vardct_decodes_look_like_photographs(oracle_backend):
    """A semantic pin of the oracle's VarDCT restatement that does not come from the oracle itself: lenna.jxl and bbb.jxl
    are photographic / rendered pictures, so a correct decode is smooth across the 8x8 varblock grid. A wrong dequantiser,
    chroma-from-luma, LLF, inverse transform, Gaborish, EPF or XYB leaves block edges, ringing or colour casts that these
    statistics catch: (1) the mean absolute step across 8-pixel cell borders is no larger than the step inside cells,
    (2) neighbouring pixels are highly correlated, (3) nearly all samples lie inside the displayable range, (4) the picture
    is not flat, and (5) the three channels are positively correlated (a luminance picture with chroma, not noise)."""
    for name in ("lenna", "bbb"):
        _, img = decode(name, oracle_backend)
        buf = io.BytesIO()
        PNGWriter(img).write(buf)
        px, bd = read_png(buf.getvalue())
        a = px[..., :3].astype(np.float64) / (255.0 if bd == 8 else 65535.0)
        g = a.mean(axis=2)
        dx = np.abs(np.diff(g, axis=1))
        cols = np.arange(dx.shape[1])
        on_border = (cols % 8) == 7          # step from column 8k+7 to 8k+8
        ratio_x = dx[:, on_border].mean() / dx[:, ~on_border].mean()
        dy = np.abs(np.diff(g, axis=0))
        rows = np.arange(dy.shape[0])
        ratio_y = dy[(rows % 8) == 7].mean() / dy[(rows % 8) != 7].mean()
        assert 0.8 < ratio_x < 1.15 and 0.8 < ratio_y < 1.15, (name, ratio_x, ratio_y)   # no blocking at the cell grid
        cx = np.corrcoef(g[:, :-1].ravel(), g[:, 1:].ravel())[0, 1]
        cy = np.corrcoef(g[:-1].ravel(), g[1:].ravel())[0, 1]
        assert cx > 0.9 and cy > 0.9, (name, cx, cy)
        assert ((px == 0) | (px == (255 if bd == 8 else 65535))).mean() < 0.25, name     # not clipped away
        assert g.std() > 0.05, name
        rg = np.corrcoef(a[..., 0].ravel(), a[..., 1].ravel())[0, 1]
        gb = np.corrcoef(a[..., 1].ravel(), a[..., 2].ravel())[0, 1]
        assert rg > 0.5 and gb > 0.5, (name, rg, gb)


def test_modular_decodes_are_structured_images(oracle_backend):
    """Semantic pins of the integer path on the reference's own sample files. patches-lossless.jxl is a lossless screenshot
    and art.jxl a JXL-art tree: a correct decode has a few hundred distinct colours (326 and 71), while any slip in the
    predictors, RCT, palette or Squeeze arithmetic smears them into tens of thousands. quilt.jxl (frame-level Squeeze, 16
    steps), blendmodes_5.jxl and wb-rainbow.jxl are continuous-tone: neighbouring pixels correlate strongly."""
    def pixels(name):
        _, img = decode(name, oracle_backend)
        buf = io.BytesIO()
        PNGWriter(img).write(buf)
        return read_png(buf.getvalue())[0]
    for name, limit in (("patches-lossless", 1000), ("art", 200)):
        px = pixels(name)
        assert len(np.unique(px.reshape(-1, px.shape[-1]), axis=0)) < limit, name
    for name, floor in (("quilt", 0.7), ("blendmodes_5", 0.95), ("wb-rainbow", 0.85)):
        g = pixels(name)[..., :3].astype(np.float64).mean(axis=2)
        cx = np.corrcoef(g[:, :-1].ravel(), g[:, 1:].ravel())[0, 1]
        cy = np.corrcoef(g[:-1].ravel(), g[1:].ravel())[0, 1]
        assert cx > floor and cy > floor, (name, cx, cy)


CRCS = {}


def pixel_crc(img):
    c = 0
    for b in img.buffer:
        a = np.ascontiguousarray(b)
        if a.dtype == np.float32:
            a = np.where(np.isnan(a), np.float32(0), a)
        c = zlib.crc32(a.tobytes(), c)
    return c


@pytest.mark.parametrize("name", DECODABLE)
def test_decode_sample_oracle_backend(oracle_backend, name):
    dec, img = decode(name, oracle_backend)
    w, h = EXPECT_HEADERS[name][:2]
    if EXPECT_HEADERS[name][5] > 4:
        w, h = h, w
    assert (img.getWidth(), img.getHeight()) == (w, h)
    for b in img.buffer:
        assert b.shape == (h, w)
        if b.dtype == np.float32:
            assert np.isfinite(b).all()
    fix = os.path.join(SAMPLES, "pixel_crc.txt")
    known = dict(l.split() for l in open(fix)) if os.path.exists(fix) else {}
    crc = "%08x" % pixel_crc(img)
    if name in known:
        assert known[name] == crc, "decoded pixels of %s changed" % name
    else:  # first run in the build container records the value (committed afterwards)
        with open(fix, "a") as f:
            f.write("%s %s\n" % (name, crc))
    if name in ("lenna", "bbb"):  # photographic content: natural-image statistics, no blocking garbage
        buf = io.BytesIO()
        PNGWriter(img).write(buf)
        px, _ = read_png(buf.getvalue())
        g = px.astype(np.float32).mean(-1)
        assert 40 < g.mean() < 215 and g.std() > 20
        # neighbouring pixels are strongly correlated in a correctly reconstructed photo
        assert np.corrcoef(g[:, :-1].ravel(), g[:, 1:].ravel())[0, 1] > 0.97


def test_wb_rainbow_exercises_every_feature_stage(oracle_backend):
    """352 bytes: 5 modular frames with 2x upsampling, noise, two splines, cropped frames blended with modes ADD / BLEND
    onto an RGBA canvas -- rows f3 / f4 on a real bitstream"""
    dec, img = decode("wb-rainbow", oracle_backend)
    assert len(dec.stats) == 5 and img.hasAlpha() and (img.getWidth(), img.getHeight()) == (2048, 1152)
    rgb = np.stack(img.buffer[:3])
    assert rgb.dtype == np.float32 and 0.2 < float(rgb.mean()) < 0.8
    # the rainbow: every hue sextant is present among saturated pixels
    mx, mn = rgb.max(0), rgb.min(0)
    sat = (mx - mn) > 0.5
    arg = rgb.argmax(0)[sat]
    assert all((arg == c).mean() > 0.1 for c in range(3))


# ---- device vs oracle on real files (GPU) --------------------------------------------------------------------
@pytest.fixture(scope="module")
def device_backend():
    from jxlatte_amd.decoder import DeviceBackend
    b = DeviceBackend(0)
    yield b
    b.close()


@pytest.mark.gpu
@pytest.mark.parametrize("name", DECODABLE)
def test_device_decode_matches_oracle_decode(device_backend, oracle_backend, name):
    _, got = decode(name, device_backend)
    _, exp = decode(name, oracle_backend)
    assert len(got.buffer) == len(exp.buffer)
    for c, (a, b) in enumerate(zip(got.buffer, exp.buffer)):
        assert_bits_equal(a, b, "%s channel %d" % (name, c))
    ga, gb = io.BytesIO(), io.BytesIO()
    PNGWriter(got).write(ga)
    PNGWriter(exp).write(gb)
    pa, _ = read_png(ga.getvalue())
    pb, _ = read_png(gb.getvalue())
    assert np.array_equal(pa, pb)  # the sRGB transfer + quantisation of the device equals the oracle's for every input (r3 tables)


# ---- large reference samples (not committed: tests/golden/samples_large/ is git-ignored but travels with gpurun) -------
LARGE = os.path.join(os.path.dirname(SAMPLES), "samples_large")


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["sollevante-hdr", "george-tiled", "ants"])
def test_device_decode_matches_oracle_decode_large(device_backend, oracle_backend, name):
    """4K HDR VarDCT (135 groups, 4 LF groups, BT.2100 PQ), a 135-frame tiled 4K image blended onto one canvas, and an 8 MP
    JPEG-recompressed 4:2:0 YCbCr frame (chroma-subsampled channels, RAW quant tables)"""
    p = os.path.join(LARGE, name + ".jxl")
    if not os.path.exists(p):
        pytest.skip("large sample not present (kept out of the repository)")
    got = JXLDecoder(p, backend=device_backend).decode()
    exp = JXLDecoder(p, backend=oracle_backend).decode()
    assert (got.getWidth(), got.getHeight()) == ((3264, 2448) if name == "ants" else (3840, 2160))
    for c, (a, b) in enumerate(zip(got.buffer, exp.buffer)):
        assert_bits_equal(a, b, "%s channel %d" % (name, c))


def test_lf_from_lf_frame_region_and_cast():
    """USE_LF_FRAME (LFCoefficients.java:44-57): an LF group's dequantised LF is a window of the planes the LF frame left
    behind, origin (lfg << 8) for every channel, integer planes cast as ImageBuffer.castToFloat does. No sample bitstream
    uses LF frames and no encoder exists here: this pins the host-side assembly only."""
    from jxlatte_amd.decoder import lf_from_lf_frame
    rng = np.random.default_rng(2)
    planes = [rng.standard_normal((300, 520)).astype(np.float32) for _ in range(3)]
    got = lf_from_lf_frame(planes, 1, 1, 44, 200, [0, 0, 0], [0, 0, 0], 8)
    for c in range(3):
        assert got[c].dtype == np.float32 and np.array_equal(got[c], planes[c][256:300, 256:456])
    sub = lf_from_lf_frame(planes, 0, 1, 40, 64, [1, 0, 1], [1, 0, 0], 8)   # 4:2:0-like shifts: smaller windows, same origin
    assert sub[0].shape == (20, 32) and sub[1].shape == (40, 64) and sub[2].shape == (20, 64)
    assert np.array_equal(sub[0], planes[0][0:20, 256:288])
    ints = [rng.integers(0, 1024, (260, 260)).astype(np.int32) for _ in range(3)]
    gi = lf_from_lf_frame(ints, 0, 0, 8, 8, [0, 0, 0], [0, 0, 0], 10)
    assert np.array_equal(gi[1], (ints[1][:8, :8].astype(np.float32) * (np.float32(1) / np.float32(1023))).astype(np.float32))


@pytest.mark.gpu
def test_wb_rainbow_colour_planes_cross_the_bus_once_each_way(device_backend):
    """row f4 chained: frame 0 (2x upsampling + noise) uploads its Modular colour planes once and downloads the result once;
    the other frames have host stages only (splines, as in the reference; the image is not XYB-encoded): nothing moves"""
    dec, _ = decode("wb-rainbow", device_backend)
    assert [s["plane_moves"] for s in dec.stats] == [["h2d", "d2h"], [], [], [], []]


def test_vardct_lf_frame_keeps_its_xyb_planes(oracle_backend):
    """a VarDCT frame with lf_level > 0 is stored in lfBuffer BEFORE the colour transform
    (JXLCodestreamDecoder.java:615-617) and read back as XYB LF coefficients (LFCoefficients.java:44-57): the inverse XYB must not
    be fused into it. lenna.jxl's frame is presented to the decode loop as an LF frame of level 1."""
    from jxlatte_amd import decoder as dmod
    dec = JXLDecoder(path("lenna"), backend=oracle_backend)
    fused, planes_seen = [], []
    orig_vf, orig_next = dec._vardct_frame, dec.fe.next_frame

    def spy_vf(fr, fuse_xyb, keep=None):
        fused.append(bool(fuse_xyb))
        out = orig_vf(fr, fuse_xyb, keep)
        planes_seen.append([np.array(p, copy=True) for p in out])
        return out

    def lf_next(*a):
        fr = orig_next(*a)
        if fr is not None:
            fr.lf_level = 1
            fr.type = dmod.LF_FRAME
        return fr
    dec._vardct_frame, dec.fe.next_frame = spy_vf, lf_next
    try:
        dec.decode()
    except Exception:
        pass  # a stream made of one LF frame produces no image: only the LF buffer matters here
    assert fused == [False]
    assert dec.lfBuffer[0] is not None
    for c in range(3):
        assert_bits_equal(dec.lfBuffer[0][c], planes_seen[0][c], "lfBuffer plane %d" % c)
    # and those planes are XYB, not linear RGB: the same frame decoded normally has the colour transform applied
    ref = JXLDecoder(path("lenna"), backend=oracle_backend)
    seen = []
    rvf = ref._vardct_frame
    ref._vardct_frame = lambda fr, fuse, keep=None: (seen.append(bool(fuse)), rvf(fr, fuse, keep))[1]
    ref.decode()
    assert seen == [True]

"""Synthetic inputs generated *at the boundary* (SURVEY.md section 8(d)): what the Java host
would hand over after entropy decoding -- quantised HF coefficients, LF planes, varblock
maps, CfL / sharpness side info -- for VarDCT frames, and squeezed residual channels for
Modular frames. No JXL encoder exists in this environment, so seeds + this generator define
the workloads (C2..C5 of BASELINE.json). numpy `default_rng(seed)` only.

The varblock tiling follows the placement rule of HFMetadata.placeBlock
(J/frame/vardct/HFMetadata.java:93-119): blocks are listed per LF group in raster order of
their top-left cell and never cross a 256x256 group.
"""
import math

import numpy as np

from . import abi, hfglobal

F = np.float32

# area shares of SURVEY 8(d) "C3 4K VarDCT"
MIX_DEFAULT = {
    "DCT8": 0.40, "DCT16": 0.15, "DCT32": 0.10, "DCT16_8": 0.05, "DCT8_16": 0.05,
    "DCT32_8": 0.02, "DCT8_32": 0.02, "DCT32_16": 0.02, "DCT16_32": 0.02, "DCT64": 0.05,
    "DCT4": 0.015, "DCT4_8": 0.015, "DCT8_4": 0.015, "DCT2": 0.015, "HORNUSS": 0.012,
    "AFV0": 0.012, "AFV1": 0.012, "AFV2": 0.012, "AFV3": 0.012,
}
MIX_DCT8 = {"DCT8": 1.0}
MIX_LARGE = {
    "DCT8": 0.20, "DCT16": 0.10, "DCT32": 0.10, "DCT64": 0.15, "DCT64_32": 0.05, "DCT32_64": 0.05,
    "DCT128": 0.10, "DCT128_64": 0.05, "DCT64_128": 0.05, "DCT256": 0.10, "DCT256_128": 0.025,
    "DCT128_256": 0.025,
}
MIX_ALL = {t[0]: 1.0 / 27 for t in abi.TRANSFORM_TYPES}
MIXES = {"default": MIX_DEFAULT, "dct8": MIX_DCT8, "large": MIX_LARGE, "all": MIX_ALL}

# OpsinInverseMatrix defaults (OpsinInverseMatrix.java:11-27)
DEFAULT_OPSIN = [11.031566901960783, -9.866943921568629, -0.16462299647058826,
                 -3.254147380392157, 4.418770392156863, -0.16462299647058826,
                 -3.6588512862745097, 2.7129230470588235, 1.9459282392156863]
DEFAULT_OPSIN_BIAS = -0.0037930732552754493
DEFAULT_QUANT_BIAS = [0.945349926692846, 0.9299455010825141, 0.9500648966626564]
DEFAULT_QBIAS_NUM = 0.145


def default_params(width, height, global_scale=2500, xqm=3, bqm=2, epf_iters=2, gab=True, xyb=True,
                   intensity_target=255.0, transfer=abi.TRANSFER_NONE, out_format=abi.OUT_F32,
                   stages=abi.STAGE_ALL, opsin_matrix=None):
    """jxl_vardct_params with the reference's header defaults (RestorationFilter.java:28-44,
    LFChannelCorrelation.java:23-27, OpsinInverseMatrix.java:11-27, FrameHeader xqm/bqm)."""
    p = abi.VarDCTParams()
    p.width, p.height, p.stages = width, height, stages
    gs = F(65536.0) / F(global_scale)
    p.scale_factor[0] = F(gs * F(math.pow(0.8, xqm - 2.0)))
    p.scale_factor[1] = gs
    p.scale_factor[2] = F(gs * F(math.pow(0.8, bqm - 2.0)))
    for c in range(3):
        p.quant_bias[c] = DEFAULT_QUANT_BIAS[c]
    p.quant_bias_numerator = DEFAULT_QBIAS_NUM
    p.base_corr_x, p.base_corr_b, p.color_factor = 0.0, 1.0, 84
    p.gab = 1 if gab else 0
    for c in range(3):
        p.gab_w1[c] = 0.115169525
        p.gab_w2[c] = 0.061248592
    p.epf_iters = epf_iters
    p.global_scale_f = gs
    quant_mul = F(0.46)
    for i in range(8):
        p.epf_sharp_lut[i] = F(F(i) / F(7.0)) * quant_mul if i < 7 else quant_mul
    for c, v in enumerate((40.0, 5.0, 3.5)):
        p.epf_channel_scale[c] = v
    p.epf_pass0_sigma_scale, p.epf_pass2_sigma_scale = 0.9, 6.5
    p.epf_border_sad_mul = F(2.0) / F(3.0)
    p.xyb = 1 if xyb else 0
    m = opsin_matrix if opsin_matrix is not None else DEFAULT_OPSIN
    for i in range(9):
        p.opsin_matrix[i] = m[i]
    for c in range(3):
        p.opsin_bias[c] = DEFAULT_OPSIN_BIAS
        # (float)Math.cbrt(opsinBias[c]) (OpsinInverseMatrix.java:83); cbrt of the float value in double
        p.cbrt_opsin_bias[c] = F(np.cbrt(np.float64(F(DEFAULT_OPSIN_BIAS))))
    p.intensity_target = intensity_target
    p.transfer, p.out_format = transfer, out_format
    return p


def bt2100_opsin_matrix():
    """A BT.2100(Rec.2020)/D65-adapted opsin matrix as OpsinInverseMatrix.getMatrix would produce
    (OpsinInverseMatrix.java:94-100): conversion(sRGB->BT.2020 primaries) x default matrix, in f32.
    The conversion matrix is derived host-side in double from the CIE xy primaries (ColorManagement)."""
    def prim_to_xyz(xy, wp):
        m = np.array([[x / y, 1.0, (1 - x - y) / y] for x, y in xy]).T
        w = np.array([wp[0] / wp[1], 1.0, (1 - wp[0] - wp[1]) / wp[1]])
        s = np.linalg.solve(m, w)
        return m * s
    d65 = (0.3127, 0.3290)
    srgb = prim_to_xyz([(0.64, 0.33), (0.30, 0.60), (0.15, 0.06)], d65)
    bt2020 = prim_to_xyz([(0.708, 0.292), (0.170, 0.797), (0.131, 0.046)], d65)
    conv = (np.linalg.inv(bt2020) @ srgb).astype(F)
    base = np.array(DEFAULT_OPSIN, F).reshape(3, 3)
    out = np.zeros((3, 3), F)
    for y in range(3):
        for x in range(3):
            acc = F(0)
            for k in range(3):
                acc = F(acc + F(conv[y, k] * base[k, x]))
            out[y, x] = acc
    return [float(v) for v in out.ravel()]


def _draw_tiling_firstfit(rng, bh, bw, mix):
    """first-fit in raster order without alignment (exercises varblocks straddling 64x64 CfL tiles)."""
    names = list(mix.keys())
    types = np.array([abi.TT_BY_NAME[n] for n in names])
    dims = np.array([(abi.TRANSFORM_TYPES[t][5] // 8, abi.TRANSFORM_TYPES[t][6] // 8) for t in types])
    share = np.array([mix[n] for n in names], np.float64)
    prob = share / np.sqrt(dims[:, 0] * dims[:, 1])
    prob /= prob.sum()
    sel = np.full((bh, bw), 255, np.uint8)
    blocks = []
    draws = rng.choice(len(names), size=bh * bw, p=prob)
    di = 0
    for y in range(bh):
        row = sel[y]
        x = 0
        while x < bw:
            if row[x] != 255:
                x += 1
                continue
            k = draws[di]
            di += 1
            t, (dh, dw) = types[k], dims[k]
            gy0, gx0 = (y // 32) * 32, (x // 32) * 32
            ok = (y + dh <= min(gy0 + 32, bh)) and (x + dw <= min(gx0 + 32, bw))
            if ok:
                ok = bool(np.all(sel[y:y + dh, x:x + dw] == 255))
            if not ok:
                t, dh, dw = 0, 1, 1
            sel[y:y + dh, x:x + dw] = t
            blocks.append((y, x, int(t)))
            x += dw
    return sel, blocks


def _draw_tiling(rng, bh, bw, mix, aligned=True):
    """Cover the bh x bw cell grid with varblocks so that each type's AREA share approximates `mix`.
    aligned=True: every block sits at a multiple of its own size (what libjxl emits); largest types are
    placed first on their aligned grids with the probability that yields their share, the remaining cells
    go to the 8x8-footprint types. Returns (type map u8[bh][bw], list of (y, x, type))."""
    if not aligned:
        return _draw_tiling_firstfit(rng, bh, bw, mix)
    sel = np.full((bh, bw), 255, np.uint8)
    free = np.ones((bh, bw), bool)
    total = float(bh * bw)
    blocks = []
    big = [(n, abi.TT_BY_NAME[n]) for n in mix if abi.TRANSFORM_TYPES[abi.TT_BY_NAME[n]][5] * abi.TRANSFORM_TYPES[abi.TT_BY_NAME[n]][6] > 64]
    big.sort(key=lambda nt: -(abi.TRANSFORM_TYPES[nt[1]][5] * abi.TRANSFORM_TYPES[nt[1]][6]))
    for name, t in big:
        dh, dw = abi.TRANSFORM_TYPES[t][5] // 8, abi.TRANSFORM_TYPES[t][6] // 8
        nyb, nxb = bh // dh, bw // dw
        if nyb == 0 or nxb == 0:
            continue
        fb = free[:nyb * dh, :nxb * dw].reshape(nyb, dh, nxb, dw).all(axis=(1, 3))
        avail = fb.sum() * dh * dw / total
        if avail <= 0:
            continue
        p = min(1.0, mix[name] / avail)
        pick = fb & (rng.random((nyb, nxb)) < p)
        ys, xs = np.nonzero(pick)
        for y, x in zip(ys.tolist(), xs.tolist()):
            sel[y * dh:(y + 1) * dh, x * dw:(x + 1) * dw] = t
            free[y * dh:(y + 1) * dh, x * dw:(x + 1) * dw] = False
            blocks.append((y * dh, x * dw, t))
    small = [(n, abi.TT_BY_NAME[n]) for n in mix if abi.TRANSFORM_TYPES[abi.TT_BY_NAME[n]][5] * abi.TRANSFORM_TYPES[abi.TT_BY_NAME[n]][6] == 64]
    if not small:
        small = [("DCT8", 0)]
    sp = np.array([mix.get(n, 1.0) for n, _ in small], np.float64)
    sp /= sp.sum()
    ys, xs = np.nonzero(free)
    pick = rng.choice(len(small), size=ys.size, p=sp)
    st = np.array([t for _, t in small], np.uint8)[pick]
    sel[ys, xs] = st
    blocks.extend(zip(ys.tolist(), xs.tolist(), st.tolist()))
    return sel, blocks


def make_vardct_frame(width, height, seed=1234, mix="default", aligned=True, params=None, nonzero_p=0.15,
                      coeff_scale=6.0, **param_kw):
    """Synthetic VarDCT frame at the boundary. width/height = padded frame size (multiples of 8).

    Returns a dict: params (abi.VarDCTParams), weights (f32 flat), woffs (i32[51]), lfgroups (list of dicts in
    jxl_lfgroup_desc shape), coeff (i32 [3][H][W], pass-summed quantizedCoeffs in frame coordinates), plus the
    frame-level maps used to build them (for tests).
    """
    assert width % 8 == 0 and height % 8 == 0
    rng = np.random.default_rng(seed)
    if isinstance(mix, str) and "=" in mix:  # explicit shares: "DCT8=0.5+DCT16=0.5"
        mix = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in mix.split("+")}
    mixd = MIXES[mix] if isinstance(mix, str) else mix
    bh, bw = height // 8, width // 8
    sel, blocks = _draw_tiling(rng, bh, bw, mixd, aligned)
    nblk = len(blocks)
    by = np.array([b[0] for b in blocks], np.int32)
    bx = np.array([b[1] for b in blocks], np.int32)
    bt = np.array([b[2] for b in blocks], np.int32)
    ph = np.array([abi.TRANSFORM_TYPES[t][5] for t in bt], np.int32)
    pw = np.array([abi.TRANSFORM_TYPES[t][6] for t in bt], np.int32)

    # cell-level maps of block origin/size/hfMul
    blk_mul = rng.integers(1, 9, size=nblk).astype(np.int32)
    cell_oy = np.zeros((bh, bw), np.int32)
    cell_ox = np.zeros((bh, bw), np.int32)
    cell_h = np.zeros((bh, bw), np.int32)
    cell_w = np.zeros((bh, bw), np.int32)
    hf_mul = np.zeros((bh, bw), np.int32)
    for i in range(nblk):
        y, x, dh, dw = by[i], bx[i], ph[i] // 8, pw[i] // 8
        cell_oy[y:y + dh, x:x + dw] = y
        cell_ox[y:y + dh, x:x + dw] = x
        cell_h[y:y + dh, x:x + dw] = ph[i]
        cell_w[y:y + dh, x:x + dw] = pw[i]
        hf_mul[y:y + dh, x:x + dw] = blk_mul[i]
    sharpness = rng.integers(0, 8, size=(bh, bw)).astype(np.int32)
    th, tw = (bh + 7) // 8, (bw + 7) // 8
    x_from_y = rng.integers(-8, 9, size=(th, tw)).astype(np.int32)
    b_from_y = rng.integers(-8, 9, size=(th, tw)).astype(np.int32)

    # LF: smooth field, 5x5 box-filtered N(0,1) * 0.1 per channel
    lf = np.zeros((3, bh, bw), F)
    for c in range(3):
        n = rng.standard_normal((bh + 4, bw + 4))
        cs = np.cumsum(np.cumsum(np.pad(n, ((1, 0), (1, 0))), 0), 1)
        box = (cs[5:, 5:] - cs[:-5, 5:] - cs[5:, :-5] + cs[:-5, :-5]) / 25.0
        lf[c] = (0.1 * (c == 1) + 0.02 * (c != 1) + 0.1 * box[:bh, :bw]).astype(F)

    # quantised HF coefficients: zero with p = 1 - nonzero_p, else round(Laplace(0, scale/(1+4r))), r = radial
    # frequency normalised to the block size; LLF corner positions stay 0 (the decoder never writes them)
    up = lambda m: np.repeat(np.repeat(m, 8, 0), 8, 1)
    yy = (np.arange(height, dtype=np.int32)[:, None] - up(cell_oy) * 8).astype(F)
    xx = (np.arange(width, dtype=np.int32)[None, :] - up(cell_ox) * 8).astype(F)
    hh, ww = up(cell_h).astype(F), up(cell_w).astype(F)
    llf = (yy * 8 < hh) & (xx * 8 < ww)
    yy /= hh
    xx /= ww
    scale = (F(coeff_scale) / (F(1.0) + F(4.0) * np.sqrt(yy * yy + xx * xx))).ravel()
    del yy, xx, hh, ww
    notllf = ~llf.ravel()
    coeff = np.zeros((3, height, width), np.int32)
    for c in range(3):
        keep = (rng.random(height * width, dtype=F) < F(nonzero_p)) & notllf
        idx = np.flatnonzero(keep)
        lap = rng.laplace(0.0, 1.0, size=idx.size) * scale[idx] * (1.0 if c == 1 else 0.5)
        coeff[c].ravel()[idx] = np.rint(lap).astype(np.int32)

    weights, woffs = hfglobal.default_weights()
    p = params if params is not None else default_params(width, height, **param_kw)
    p.width, p.height = width, height

    # split frame-level maps into LF groups (2048 px = 256 cells)
    lfgroups = []
    lrs, lcs = (bh + 255) // 256, (bw + 255) // 256
    order = np.lexsort((bx, by))  # raster order of block origins
    for ly in range(lrs):
        for lx in range(lcs):
            y0, x0 = ly * 256, lx * 256
            y1, x1 = min(y0 + 256, bh), min(x0 + 256, bw)
            inside = order[(by[order] >= y0) & (by[order] < y1) & (bx[order] >= x0) & (bx[order] < x1)]
            byx = np.stack([by[inside] - y0, bx[inside] - x0], 1).astype(np.int32)
            g = dict(
                lfg_y=ly, lfg_x=lx,
                dct_select=np.ascontiguousarray(sel[y0:y1, x0:x1]),
                hf_mul=np.ascontiguousarray(hf_mul[y0:y1, x0:x1]),
                sharpness=np.ascontiguousarray(sharpness[y0:y1, x0:x1]),
                x_from_y=np.ascontiguousarray(x_from_y[y0 // 8:(y1 + 7) // 8, x0 // 8:(x1 + 7) // 8]),
                b_from_y=np.ascontiguousarray(b_from_y[y0 // 8:(y1 + 7) // 8, x0 // 8:(x1 + 7) // 8]),
                block_yx=np.ascontiguousarray(byx),
                lf=[np.ascontiguousarray(lf[c, y0:y1, x0:x1]) for c in range(3)],
            )
            lfgroups.append(g)
    return dict(params=p, weights=weights, woffs=woffs, lfgroups=lfgroups, coeff=coeff,
                width=width, height=height, n_blocks=nblk, dct_select=sel, hf_mul=hf_mul, sharpness=sharpness,
                block_types=bt, mix=mix, seed=seed)


def channel_planes(frame):
    """views of the coefficient planes in each channel's own geometry: for a chroma-subsampled frame channel c occupies
    the first (H >> sy) * (W >> sx) samples of coeff[c], row stride W >> sx (the layout the oracle reads)"""
    W, H = frame["width"], frame["height"]
    p = abi.VarDCTParams.from_buffer_copy(frame["params"])
    out = []
    for c in range(3):
        h, w = H >> p.jpeg_upsampling_y[c], W >> p.jpeg_upsampling_x[c]
        out.append(frame["coeff"][c].reshape(-1)[:h * w].reshape(h, w))
    return out, list(p.jpeg_upsampling_y), list(p.jpeg_upsampling_x)


def group_view(frame, group):
    """per-group coefficient planes as the reference holds them (HFCoefficients.quantizedCoeffs):
    returns list of 3 contiguous int32 arrays [gh >> sy][gw >> sx]."""
    W, H = frame["width"], frame["height"]
    grs = (W + 255) // 256
    gy, gx = divmod(group, grs)
    y0, x0 = gy * 256, gx * 256
    planes, sy, sx = channel_planes(frame)
    return [np.ascontiguousarray(planes[c][y0 >> sy[c]:min(y0 + 256, H) >> sy[c], x0 >> sx[c]:min(x0 + 256, W) >> sx[c]])
            for c in range(3)]


def make_subsampled(frame, sy, sx):
    """Turn an all-DCT8 synthetic frame into a chroma-subsampled one (JPEG-recompression geometry, FrameHeader
    jpegUpsamplingY/X = sy/sx per channel): the varblocks of channel c that survive are those on even cells, and their
    coefficients / LF samples move onto the channel's own (H >> sy) x (W >> sx) grid. Gab / EPF keep running on the
    full-size planes after Frame.invertSubsampling."""
    W, H = frame["width"], frame["height"]
    assert W % 16 == 0 and H % 16 == 0
    p = abi.VarDCTParams.from_buffer_copy(frame["params"])
    out = dict(frame)
    coeff = np.zeros_like(frame["coeff"])
    for c in range(3):
        p.jpeg_upsampling_y[c], p.jpeg_upsampling_x[c] = sy[c], sx[c]
        h, w = H >> sy[c], W >> sx[c]
        full = frame["coeff"][c].reshape(H // 8, 8, W // 8, 8)
        sub = full[::1 << sy[c], :, ::1 << sx[c], :]  # blocks on aligned cells, in their new positions
        coeff[c].reshape(-1)[:h * w] = np.ascontiguousarray(sub).reshape(h, w).reshape(-1)
    out["coeff"] = coeff
    out["params"] = p
    groups = []
    for g in frame["lfgroups"]:
        assert (g["dct_select"] == 0).all(), "make_subsampled needs an all-DCT8 tiling"
        g2 = dict(g)
        g2["lf"] = [np.ascontiguousarray(g["lf"][c][::1 << sy[c], ::1 << sx[c]]) for c in range(3)]
        groups.append(g2)
    out["lfgroups"] = groups
    return out


def num_groups(frame):
    return ((frame["width"] + 255) // 256) * ((frame["height"] + 255) // 256)


def type_histogram(frame):
    """area share per transform type, for reporting next to every number."""
    out = {}
    total = float(frame["width"] * frame["height"])
    for t in np.unique(frame["block_types"]):
        n = int(np.sum(frame["block_types"] == t))
        ph, pw = abi.tt_pixel_size(int(t))
        out[abi.TT_NAME[int(t)]] = round(n * ph * pw / total, 4)
    return out


# ---- Modular -------------------------------------------------------------------------------------

def default_squeeze_params(shapes, nb_meta=0):
    """ModularStream.java:110-131 (host logic, restated; the C-ABI exports the same rule)."""
    sp = []
    first = nb_meta
    count = len(shapes) - first
    if count <= 0:
        return sp
    h, w = shapes[0]
    if count > 2 and shapes[first + 1] == (h, w):
        sp.append((1, 0, first + 1, 2))
        sp.append((0, 0, first + 1, 2))
    if h >= w and h > 8:
        sp.append((0, 1, first, count))
        h = (h + 1) // 2
    while w > 8 or h > 8:
        if w > 8:
            sp.append((1, 1, first, count))
            w = (w + 1) // 2
        if h > 8:
            sp.append((0, 1, first, count))
            h = (h + 1) // 2
    return sp


def squeezed_shapes(shapes, sp):
    """forward channel-list surgery of the ModularStream ctor (ModularStream.java:137-167)."""
    shapes = [tuple(s) for s in shapes]
    for (horiz, in_place, begin, num) in sp:
        end = begin + num - 1
        offset = end + 1 if in_place else len(shapes)
        for k in range(begin, end + 1):
            r = offset + k - begin
            h, w = shapes[k]
            if horiz:
                shapes[k] = (h, (w + 1) // 2)
                res = (h, w // 2)
            else:
                shapes[k] = ((h + 1) // 2, w)
                res = (h // 2, w)
            shapes.insert(r, res)
    return shapes


def make_modular_frame(width, height, channels=3, seed=7, sp=None, res_scale=4.0):
    """Encoded (squeezed) channel list for a width x height, `channels`-channel image with the default
    squeeze plan: coarsest averages ~ U{0..255}, residual channels ~ round(Laplace(0, res_scale)).
    Returns dict(chans=[int32 arrays], sp=[...], shapes=[(h, w)...])."""
    rng = np.random.default_rng(seed)
    img_shapes = [(height, width)] * channels
    if sp is None:
        sp = default_squeeze_params(img_shapes)
    enc = squeezed_shapes(img_shapes, sp)
    chans = []
    for i, (h, w) in enumerate(enc):
        if i < channels:
            a = rng.integers(0, 256, size=(h, w)).astype(np.int32)
        else:
            # (clipped BEFORE the cast: a float outside int32 converts to a platform-defined value, and the fuzz's wide res_scale reaches it)
            a = np.clip(np.rint(rng.laplace(0.0, res_scale, size=(h, w))), -2147483648.0, 2147483647.0).astype(np.int64).astype(np.int32)
        chans.append(np.ascontiguousarray(a))
    return dict(chans=chans, sp=sp, shapes=enc, width=width, height=height, channels=channels, seed=seed)

"""JXLDecoder / JXLImage / PNGWriter: the reference's user-facing API (J/JXLDecoder.java, J/JXLCodestreamDecoder.java,
J/JXLImage.java, J/io/PNGWriter.java) on top of the C++ bitstream front-end (jxlatte_amd.frontend, row f2) and the
device library (jxlatte_amd._lib / host, the hot path).

    from jxlatte_amd.decoder import JXLDecoder, PNGWriter
    image = JXLDecoder("in.jxl").decode()
    PNGWriter(image).write(open("out.png", "wb"))

Every pixel operation goes through a *backend* object. The product backend is DeviceBackend (HIP kernels through the
C-ABI); it fails loudly without a GPU. Tests construct the decoder with the oracle-backed backend from
oracle/pybackend.py to exercise this host logic on CPU-only machines -- the product package never imports it.
"""
import ctypes as C
import math
import struct
import zlib

import numpy as np

from . import abi, frontend, hfglobal

F = np.float32

# ---- FrameFlags / ColorFlags constants (J/frame/FrameFlags.java, J/color/ColorFlags.java) ------------------------
REGULAR_FRAME, LF_FRAME, REFERENCE_ONLY, SKIP_PROGRESSIVE = 0, 1, 2, 3
VARDCT, MODULAR = 0, 1
FLAG_NOISE, FLAG_PATCHES, FLAG_SPLINES, FLAG_USE_LF_FRAME, FLAG_SKIP_ADAPTIVE_LF = 1, 2, 16, 32, 128
CE_RGB, CE_GRAY, CE_XYB = 0, 1, 2
TF_BT709, TF_UNKNOWN, TF_LINEAR, TF_SRGB, TF_PQ, TF_DCI, TF_HLG = [(1 << 24) + v for v in (1, 2, 8, 13, 16, 17, 18)]
PRI_SRGB = np.array([0.639998686, 0.330010138, 0.300003784, 0.600003357, 0.150002046, 0.059997204], F)
PRI_BT2100 = np.array([0.708, 0.292, 0.170, 0.797, 0.131, 0.046], F)
PRI_P3 = np.array([0.680, 0.320, 0.265, 0.690, 0.150, 0.060], F)
WP_D65 = np.array([0.3127, 0.3290], F)
WP_D50 = np.array([0.34567, 0.34567], F)
PEAK_DETECT_AUTO, PEAK_DETECT_ON, PEAK_DETECT_OFF = -1, 1, 0


class InvalidBitstreamException(IOError):
    pass


class UnsupportedOperationException(RuntimeError):
    pass


# ---- colour management (J/color/ColorManagement.java, J/util/MathHelper.java matrix helpers), float32 ------------
def _xy_matches(a, b):
    return abs(F(a[0]) - F(b[0])) + abs(F(a[1]) - F(b[1])) < 1e-4


def _prim_matches(a, b):
    return all(_xy_matches(a[2 * i:2 * i + 2], b[2 * i:2 * i + 2]) for i in range(3))


def _mat_mul(left, right):
    """MathHelper.matrixMultiply: result[y] = left[y] . right with the row-vector accumulation order of :257-269"""
    if left is None:
        return right
    if right is None:
        return left
    out = np.zeros((3, 3), F)
    for y in range(3):
        for k in range(3):
            for x in range(3):
                out[y, x] = F(out[y, x] + F(left[y, k] * right[k, x]))
    return out


def _mat_vec(m, v):
    out = np.zeros(3, F)
    for y in range(3):
        for x in range(3):
            out[y] = F(out[y] + F(m[y, x] * v[x]))
    return out


def _invert3(m):
    """MathHelper.invertMatrix3x3 (:296-322)"""
    det = F(0)
    for c in range(3):
        c1, c2 = (c + 1) % 3, (c + 2) % 3
        det = F(det + F(F(F(m[c, 0] * m[c1, 1]) * m[c2, 2]) - F(F(m[c, 0] * m[c1, 2]) * m[c2, 1])))
    if det == 0:
        return None
    inv_det = F(F(1) / det)
    out = np.zeros((3, 3), F)
    for x in range(3):
        for y in range(3):
            x1, x2, y1, y2 = (x + 1) % 3, (x + 2) % 3, (y + 1) % 3, (y + 2) % 3
            out[y, x] = F(F(F(m[x1, y1] * m[x2, y2]) - F(m[x2, y1] * m[x1, y2])) * inv_det)
    return out


_BRADFORD = np.array([[0.8951, 0.2664, -0.1614], [-0.7502, 1.7135, 0.0367], [0.0389, -0.0685, 1.0296]], F)
_BRADFORD_INV = _invert3(_BRADFORD)


def _xyz(xy):
    if xy[0] < 0 or xy[0] > 1 or xy[1] <= 0 or xy[1] > 1:
        raise ValueError("chromaticity out of range")
    inv_y = F(F(1) / F(xy[1]))
    return np.array([F(F(xy[0]) * inv_y), F(1), F(F(F(F(1) - F(xy[0])) - F(xy[1])) * inv_y)], F)


def _adapt_white_point(target, current):
    target = WP_D50 if target is None else target
    current = WP_D50 if current is None else current
    lms_c = _mat_vec(_BRADFORD, _xyz(current))
    lms_t = _mat_vec(_BRADFORD, _xyz(target))
    if np.any(np.abs(lms_c) < 1e-8):
        raise ValueError("degenerate white point")
    a = np.zeros((3, 3), F)
    for i in range(3):
        a[i, i] = F(lms_t[i] / lms_c[i])
    return _mat_mul(_mat_mul(_BRADFORD_INV, a), _BRADFORD)


def _primaries_to_xyz(prim, wp):
    wp = WP_D50 if wp is None else wp
    pm = np.stack([_xyz(prim[0:2]), _xyz(prim[2:4]), _xyz(prim[4:6])]).T.astype(F).copy()
    xyz = _mat_vec(_invert3(pm), _xyz(wp))
    return _mat_mul(pm, np.diag(xyz).astype(F))


def get_conversion_matrix(target_prim, target_wp, current_prim, current_wp):
    """ColorManagement.getConversionMatrix (:141-151)"""
    if _prim_matches(target_prim, current_prim) and _xy_matches(target_wp, current_wp):
        return np.eye(3, dtype=F)
    wpc = None if _xy_matches(target_wp, current_wp) else _adapt_white_point(target_wp, current_wp)
    forward = _primaries_to_xyz(current_prim, current_wp)
    reverse = _invert3(_primaries_to_xyz(target_prim, target_wp))
    return _mat_mul(_mat_mul(reverse, wpc), forward)


def _to_linear(buf, tf):
    """TransferFunction.toLinearF for the transfer functions an un-XYB image can be tagged with (host side)"""
    b = buf.astype(F)
    if tf == TF_LINEAR:
        return b
    if tf == TF_SRGB:
        hi = np.power((b * F(0.9478672985781991) + F(0.052132701)).astype(np.float64), 2.4).astype(F)
        return np.where(b < F(0.0404482362771082), b * F(0.07739938080495357), hi).astype(F)
    d = b.astype(np.float64)
    if tf == TF_BT709:
        hi = np.power((d + 0.0992968268094429403) * 0.90967241568627260377, 2.2222222222222222222)
        return np.where(d < 0.081242858298635133011, d * 0.22222222222222222222, hi).astype(F)
    if tf == TF_PQ:
        e = np.power(d, 0.012683313515655965121)
        return np.power((e - 0.8359375) / (18.8515625 + 18.6875 * e), 6.2725880551301684533).astype(F)
    if tf == TF_HLG:
        raise UnsupportedOperationException("Not yet implemented")  # ColorManagement.java:164
    gamma = 3846154 if tf == TF_DCI else tf
    if gamma < (1 << 24):
        return np.power(d, 1e7 / gamma).astype(F)  # GammaTransferFunction
    raise ValueError("Invalid transfer function")


# ---- splines (J/frame/features/spline/Spline.java): host-side feature, as in the reference ------------------------------
_SQRT_H = F(math.sqrt(0.5))
_SQRT_F = F(math.sqrt(0.125))


def _erf(z):
    """MathHelper.erf (MathHelper.java:40-66), vectorised float32 with the reference's operation order"""
    z = z.astype(F)
    az = np.abs(z)
    t = (F(1) / (az * F(0.5) + F(1))).astype(F)
    u = t * F(0.17087277) - F(0.82215223)
    for cst, sign in ((1.48851587, 1), (1.13520398, -1), (0.27886807, 1), (0.18628806, -1), (0.09678418, 1), (0.37409196, 1),
                      (1.00002368, 1)):
        u = (t * u + F(cst)).astype(F) if sign > 0 else (t * u - F(cst)).astype(F)
    u = (t * u - F(1.26551223)).astype(F)
    big = (F(1) - t * np.exp((-(z * z) + u).astype(np.float64)).astype(F)).astype(F)
    t2 = (F(1) / (az * F(0.47047) + F(1))).astype(F)
    u2 = (t2 * ((t2 * ((t2 * F(0.7478556) - F(0.0958798)).astype(F)) + F(0.3480242)).astype(F))).astype(F)
    small = (F(1) - u2 * np.exp((-(z * z)).astype(np.float64)).astype(F)).astype(F)
    a = np.where(az > F(1e-4), big, small).astype(F)
    return np.where(z < 0, -a, a).astype(F)


def _spline_arcs(control):
    """Spline.upsampleControlPoints + computeIntermediarySamples(1.0f): list of (y, x, arcLength)"""
    cp = [(int(y), int(x)) for y, x in control]
    if len(cp) == 1:
        uy, ux = [F(cp[0][0])], [F(cp[0][1])]
    else:
        ext = [(cp[0][0] * 2 - cp[1][0], cp[0][1] * 2 - cp[1][1])] + cp + [(cp[-1][0] * 2 - cp[-2][0], cp[-1][1] * 2 - cp[-2][1])]
        n = 16 * (len(ext) - 3) + 1
        uy, ux = [F(0)] * n, [F(0)] * n
        for i in range(len(ext) - 3):
            pY = [F(ext[i + k][0]) for k in range(4)]
            pX = [F(ext[i + k][1]) for k in range(4)]
            uy[i << 4], ux[i << 4] = pY[1], pX[1]
            t = [F(0)] * 4
            dY, dX = [F(0)] * 3, [F(0)] * 3
            for k in range(3):
                dY[k], dX[k] = F(pY[k + 1] - pY[k]), F(pX[k + 1] - pX[k])
                t[k + 1] = F(t[k] + F(math.pow(float(F(F(dY[k] * dY[k]) + F(dX[k] * dX[k]))), 0.25)))
            for step in range(1, 16):
                knot = F(t[1] + F(F(F(0.0625) * F(step)) * F(t[2] - t[1])))
                aY, aX = [F(0)] * 3, [F(0)] * 3
                with np.errstate(all="ignore"):
                    for k in range(3):
                        f = F(F(knot - t[k]) / F(t[k + 1] - t[k]))
                        aY[k], aX[k] = F(F(dY[k] * f) + pY[k]), F(F(dX[k] * f) + pX[k])
                    bY, bX = [F(0)] * 2, [F(0)] * 2
                    for k in range(2):
                        f = F(F(knot - t[k]) / F(t[k + 2] - t[k]))
                        bY[k], bX[k] = F(F(F(aY[k + 1] - aY[k]) * f) + aY[k]), F(F(F(aX[k + 1] - aX[k]) * f) + aX[k])
                    f = F(F(knot - t[1]) / F(t[2] - t[1]))
                    uy[i * 16 + step] = F(F(F(bY[1] - bY[0]) * f) + bY[0])
                    ux[i * 16 + step] = F(F(F(bX[1] - bX[0]) * f) + bX[0])
        uy[-1], ux[-1] = F(cp[-1][0]), F(cp[-1][1])
    rd = F(1.0)
    cy, cx = uy[0], ux[0]
    nxt = 0
    arcs = [(cy, cx, rd)]
    while nxt < len(uy):
        py, px = cy, cx
        acc = F(0)
        while True:
            if nxt >= len(uy):
                arcs.append((py, px, acc))
                break
            ny, nx = uy[nxt], ux[nxt]
            dy, dx = F(ny - py), F(nx - px)
            to_next = F(math.sqrt(float(F(F(dy * dy) + F(dx * dx)))))
            if F(acc + to_next) >= rd:
                f = F(F(rd - acc) / to_next)
                cy, cx = F(F(dy * f) + py), F(F(dx * f) + px)
                arcs.append((cy, cx, rd))
                break
            acc = F(acc + to_next)
            py, px = ny, nx
            nxt += 1
    return arcs


def _fourier_ict(coeffs, t):
    total = F(_SQRT_H * coeffs[0])
    for i in range(1, 32):
        total = F(total + F(coeffs[i] * F(math.cos(i * (math.pi / 32.0) * (float(t) + 0.5)))))
    return total


def render_splines(buffers, splines, base_corr_x, base_corr_b, width, height):
    """Frame.renderSplines + Spline.renderSpline (Spline.java:157-201). The reference never stores the spline index
    (Spline.java:22-24), so every spline is drawn with the coefficients of spline 0; `MathHelper.max(float...)` returns the
    minimum (MathHelper.java:190-195). Both restated as they are."""
    if not splines:
        return
    s0 = splines[0]
    qa = F(F(s0["quant_adjust"]) / F(8))
    inv_qa = F(F(1) / F(F(1) + qa)) if qa >= 0 else F(F(1) - qa)
    adj = [F(F(0.005939697) * inv_qa), F(F(0.106066017) * inv_qa), F(F(0.098994949) * inv_qa), F(F(0.47135738) * inv_qa)]
    cY = [F(F(v) * adj[1]) for v in s0["coeff"][1]]
    cX = [F(F(F(v) * adj[0]) + F(F(base_corr_x) * cY[i])) for i, v in enumerate(s0["coeff"][0])]
    cB = [F(F(F(v) * adj[2]) + F(F(base_corr_b) * cY[i])) for i, v in enumerate(s0["coeff"][2])]
    cS = [F(F(v) * adj[3]) for v in s0["coeff"][3]]
    for sp in splines:
        arcs = _spline_arcs(sp["control"])
        rd = F(1.0)
        arc_len = F(F(F(len(arcs)) - F(2)) * rd + arcs[-1][2])
        if arc_len <= 0:
            continue
        for i, (ay, ax, alen) in enumerate(arcs):
            prog = min(F(1.0), F(F(F(i) * rd) / arc_len))
            t = F(F(31) * prog)
            vals = [F(_fourier_ict(c, t) * alen) for c in (cX, cY, cB)]
            sigma = _fourier_ict(cS, t)
            with np.errstate(all="ignore"):
                inv_sigma = F(F(1) / sigma)
                max_color = min(F(0.01), vals[0], vals[1], vals[2])  # MathHelper.max(float...) is a minimum
                md = F(math.sqrt(float(F(F(F(-2) * sigma) * sigma * F(F(F(math.log(0.1)) * F(3)) - max_color))))) \
                    if F(F(F(-2) * sigma) * sigma * F(F(F(math.log(0.1)) * F(3)) - max_color)) >= 0 else F(np.nan)
            if not np.isfinite(md):
                continue  # (int)(NaN + 0.5f) == 0 for both bounds of an empty-ish box; nothing sensible to draw

            def rnd(v):
                return int(np.trunc(np.clip(F(v + F(0.5)), -2**31, 2**31 - 1)))
            xb, xe = max(0, rnd(F(ax - md))), min(width - 1, rnd(F(ax + md)))
            yb, ye = max(0, rnd(F(ay - md))), min(height - 1, rnd(F(ay + md)))
            if xb > xe or yb > ye:
                continue
            ys = np.arange(yb, ye + 1, dtype=F)[:, None]
            xs = np.arange(xb, xe + 1, dtype=F)[None, :]
            dy, dx = (ys - ay).astype(F), (xs - ax).astype(F)
            dist = np.sqrt(((dy * dy).astype(F) + (dx * dx).astype(F)).astype(np.float64)).astype(F)
            with np.errstate(all="ignore"):
                fac = _erf(((F(0.5) * dist).astype(F) + _SQRT_F).astype(F) * inv_sigma)
                fac = (fac - _erf(((F(0.5) * dist).astype(F) - _SQRT_F).astype(F) * inv_sigma)).astype(F)
                for c in range(3):
                    extra = ((((F(0.25) * vals[c]) * sigma).astype(F) * fac).astype(F) * fac).astype(F)
                    buffers[c][yb:ye + 1, xb:xe + 1] = (buffers[c][yb:ye + 1, xb:xe + 1] + extra).astype(F)


# ---- backends -------------------------------------------------------------------------------------------------
def lf_from_lf_frame(lf_buffer, lfg_y, lfg_x, cells_h, cells_w, jpeg_up_y, jpeg_up_x, bits_per_sample):
    """LFCoefficients.java:44-57 (USE_LF_FRAME): the dequantised LF of LF group (lfg_y, lfg_x) is a copy out of the planes a
    preceding LF frame left in lfBuffer[] -- rows pY .. pY + size, columns from pX, with pY = lfg_y << 8, pX = lfg_x << 8 for
    every channel (the reference does not shift the origin of a subsampled channel; kept) -- after ImageBuffer.castToFloat
    (integer planes: v * (1f / maxValue)). Returns three float32 planes in X, Y, B order."""
    py, px = lfg_y << 8, lfg_x << 8
    out = []
    for c in range(3):
        b = lf_buffer[c]
        if b.dtype != np.float32:
            b = (b.astype(np.float32) * (F(1) / F((1 << bits_per_sample) - 1))).astype(np.float32)
        h, w = cells_h >> jpeg_up_y[c], cells_w >> jpeg_up_x[c]
        out.append(np.ascontiguousarray(b[py:py + h, px:px + w], np.float32))
    return out


class DeviceBackend:
    """the product backend: HIP kernels through the C-ABI (jxlatte_amd._lib / host). No CPU fallback."""

    def __init__(self, device=0):
        from . import _lib, host
        self.host = host
        self.ctx = _lib.Context(device)

    def close(self):
        self.ctx.close()

    resident = True  # vardct(keep=(h, w)) / keep_planes(planes) hand back host.ResidentPlanes

    def keep_planes(self, planes):
        return self.host.ResidentPlanes.upload(self.ctx, planes)

    def vardct(self, params, weights, woffs, lfgroups, groups, keep=None):
        fr = self.host.Frame(self.ctx, params, weights, woffs)
        for g in lfgroups:
            fr.setLFGroup(g)
            if g.get("lf_quant") is not None:
                fr.setLFGroupQuant(g["lfg_y"], g["lfg_x"], g["lf_quant"], g["scaled_dequant"], g["extra_precision"], g["x_factor_lf"],
                                   g["b_factor_lf"], g["adaptive_smoothing"])
        for pass_, grp, q in groups:
            fr.putGroup(pass_, grp, q)
        if keep is not None:  # the planes stay on the device for the stages after decodeFrame: host.ResidentPlanes
            return fr.keepPlanes(*keep)
        return fr.decodeFrame()

    def gab(self, planes, w1, w2):
        return self.host.performGabConvolution(self.ctx, planes, w1, w2)

    def epf(self, planes, iters, inv_sigma, sigma_modular, rf):
        return self.host.performEdgePreservingFilter(self.ctx, planes, iters, inv_sigma, invModularSigma=sigma_modular,
                                                     epfChannelScale=rf["channel_scale"], epfPass0SigmaScale=rf["pass0"],
                                                     epfPass2SigmaScale=rf["pass2"], epfBorderSadMul=rf["border_sad_mul"])

    def xyb(self, planes, matrix, opsin_bias, cbrt_bias, intensity_target):
        return self.host.OpsinInverseMatrix(matrix, opsin_bias, cbrt_bias).invertXYB(self.ctx, planes, intensity_target)

    def ycbcr(self, planes):
        return self.host.performColorTransformsYCbCr(self.ctx, planes)

    def squeeze(self, ins, steps, shapes):
        ms = self.host.ModularStream(self.ctx, ins, steps)
        out = ms.applyTransforms()
        assert [o.shape for o in out] == list(shapes)
        return out

    def rct(self, a, b, c, rct_type):
        return self.host.rct(self.ctx, np.stack([a, b, c]), rct_type)

    def modular_to_float(self, a, b, scale):
        return self.host.modularToFloat(self.ctx, a, b, scale)

    def chroma_upsample(self, plane, xs, ys):
        return self.host.invertSubsampling(self.ctx, plane, xs, ys)

    def upsample(self, plane, k, weights):
        return self.host.performUpsampling(self.ctx, plane, k, weights)

    def noise_init(self, h, w, seed0, group_dim, colors):
        return self.host.initializeNoise(self.ctx, h, w, seed0, group_dim, colors)

    def noise_add(self, planes, noise, lut, bcx, bcb):
        return self.host.synthesizeNoise(self.ctx, planes, noise, lut, bcx, bcb)

    def blend(self, mode, canvas, frame, ref, rect, **kw):
        return self.host.blend(self.ctx, mode, canvas, frame, ref, rect, **kw)

    def orient(self, plane, orientation):
        return self.host.transposeBuffer(self.ctx, plane, orientation)

    def transfer(self, plane, tf):
        code = {TF_PQ: abi.TRANSFER_PQ, TF_SRGB: abi.TRANSFER_SRGB}[tf]
        return self.host.transfer(self.ctx, plane, code, 0)

    def pack(self, planes, bit_depth, alpha, premultiplied, tagged, big_endian):
        return self.host.packSamples(self.ctx, planes, bit_depth, alpha=alpha, premultiplied=premultiplied, taggedDepth=tagged,
                                     bigEndian=big_endian)


# ---- JXLImage (J/JXLImage.java) ---------------------------------------------------------------------------------
class JXLImage:
    def __init__(self, buffer, info, backend):
        self.info = info
        self.backend = backend
        self.buffer = buffer  # list of 2-D arrays (int32 or float32), colour channels first
        self.height, self.width = buffer[0].shape
        self.colorEncoding = info.colour_space
        alphas = [i for i in range(info.num_extra) if info.ec_type[i] == 0]
        self.alphaIndex = alphas[0] if alphas else -1
        self.primariesXY = np.array(info.prim_xy, F)
        self.whiteXY = np.array(info.white_xy, F)
        self.taggedTransfer = info.transfer
        self.transfer_ = TF_LINEAR if info.xyb_encoded else info.transfer
        self.alphaIsPremultiplied = self.alphaIndex >= 0 and bool(info.ec_alpha_associated[self.alphaIndex])
        colors = self.getColorChannelCount()
        self.bitDepths = [info.bits_per_sample if c < colors else info.ec_bits[c - colors] for c in range(len(buffer))]
        self.has_icc = bool(info.use_icc) and not info.xyb_encoded

    def _clone(self, buffer=None):
        im = JXLImage.__new__(JXLImage)
        im.__dict__.update(self.__dict__)
        im.buffer = list(self.buffer if buffer is None else buffer)
        im.bitDepths = list(self.bitDepths)
        return im

    def getWidth(self):
        return self.width

    def getHeight(self):
        return self.height

    def getColorChannelCount(self):
        return 1 if self.colorEncoding == CE_GRAY else 3

    def getAlphaIndex(self):
        return self.alphaIndex

    def hasAlpha(self):
        return self.alphaIndex >= 0

    def isAlphaPremultiplied(self):
        return self.alphaIsPremultiplied

    def getTaggedBitDepth(self, c):
        return self.bitDepths[c]

    def getBuffer(self, copy=True):
        return [b.copy() for b in self.buffer] if copy else self.buffer

    def isHDR(self):
        if self.taggedTransfer in (TF_PQ, TF_HLG, TF_LINEAR):
            return True
        prim = np.array(self.info.prim_xy, F)
        return not _prim_matches(prim, PRI_SRGB) and not _prim_matches(prim, PRI_P3)

    def _as_float(self, c, with_depth=True):
        b = self.buffer[c]
        if b.dtype == np.float32:
            return b
        mx = ((1 << self.bitDepths[c]) - 1) if with_depth else self.bitDepths[c]
        return (b.astype(F) * F(F(1) / F(mx))).astype(F)  # ImageBuffer.castToFloat0

    def linearize(self):
        if self.transfer_ == TF_LINEAR:
            return self
        im = self._clone()
        for c in range(self.getColorChannelCount()):
            im.buffer[c] = _to_linear(self._as_float(c), self.transfer_)
        im.transfer_ = TF_LINEAR
        return im

    def _determine_peak(self):
        im = self.linearize()
        c = 0 if im.colorEncoding == CE_GRAY else 1
        # MathHelper.max(float...) returns the MINIMUM of each row (MathHelper.java:190-195); the
        # reference then takes the maximum over rows. Restated as is.
        b = im._as_float(c)
        if im.buffer[c].dtype == np.int32:
            return F(im.buffer[c].max(axis=1).max()) / F((1 << im.bitDepths[c]) - 1)
        return F(b.min(axis=1).max())

    def transfer(self, transfer, peakDetect):
        """JXLImage.transfer(int, int) (:263-286)"""
        if transfer == self.transfer_:
            return self
        im = self.linearize()
        if self.taggedTransfer == TF_PQ and peakDetect in (PEAK_DETECT_AUTO, PEAK_DETECT_ON):
            to_pq = transfer in (TF_PQ, TF_LINEAR)
            from_pq = self.transfer_ in (TF_PQ, TF_LINEAR)
            if from_pq and not to_pq:
                scale = F(F(1) / im._determine_peak())
                if scale > 1.0 or peakDetect == PEAK_DETECT_ON:
                    im = im._clone()
                    for c in range(im.getColorChannelCount()):
                        im.buffer[c] = (im._as_float(c) * scale).astype(F)
        if im is self:
            im = self._clone()
        for c in range(im.getColorChannelCount()):
            # transferInPlace casts int planes with the bit depth itself as max value (JXLImage.java:248); float planes
            # (every XYB image) are unaffected by that
            src = im._as_float(c, with_depth=False)
            if transfer in (TF_PQ, TF_SRGB):
                im.buffer[c] = self.backend.transfer(src, transfer)  # device: TransferFunction.fromLinearF
            elif transfer == TF_LINEAR:
                im.buffer[c] = src
            else:
                raise UnsupportedOperationException("output transfer function %d" % (transfer - (1 << 24)))
        im.transfer_ = transfer
        return im

    def fillColor(self):
        if self.colorEncoding != CE_GRAY:
            return self
        im = self._clone([self.buffer[0].copy(), self.buffer[0].copy()] + list(self.buffer))
        im.bitDepths = [self.bitDepths[0]] * 2 + list(self.bitDepths)
        im.colorEncoding = CE_RGB
        return im

    def toneMapLinear(self, primaries, whitePoint):
        if _prim_matches(self.primariesXY, primaries) and _xy_matches(self.whiteXY, whitePoint):
            return self
        m = get_conversion_matrix(primaries, whitePoint, self.primariesXY, self.whiteXY)
        src = [self._as_float(c) for c in range(3)]
        im = self._clone()
        for r in range(3):  # MathHelper.matrixMutliply3InPlace: (m0*a + m1*b) + m2*c
            im.buffer[r] = ((m[r, 0] * src[0] + m[r, 1] * src[1]).astype(F) + m[r, 2] * src[2]).astype(F)
        im.primariesXY, im.whiteXY = np.array(primaries, F), np.array(whitePoint, F)
        return im

    def transform(self, primaries, whitePoint, transfer, peakDetect=PEAK_DETECT_AUTO):
        """JXLImage.transform (:186-193)"""
        if _prim_matches(primaries, self.primariesXY) and _xy_matches(whitePoint, self.whiteXY):
            return self.transfer(transfer, peakDetect)
        return self.linearize().fillColor().toneMapLinear(primaries, whitePoint).transfer(transfer, peakDetect)


# ---- JXLDecoder (J/JXLDecoder.java + J/JXLCodestreamDecoder.java) ---------------------------------------------
def _tt_dims():
    return [(t[5] >> 3, t[6] >> 3) for t in abi.TRANSFORM_TYPES]


class JXLDecoder:
    def __init__(self, source, backend=None):
        if isinstance(source, (bytes, bytearray, memoryview)):
            data = bytes(source)
        else:
            with open(source, "rb") as f:
                data = f.read()
        self.backend = backend if backend is not None else DeviceBackend()
        try:
            self.fe = frontend.Frontend(data)
        except frontend.FrontendError as e:
            raise self._map(e)
        self.info = self.fe.image
        self.reference = [None] * 4
        self.lfBuffer = [None] * 5
        self.canvas = None
        self.visibleFrames = 0
        self.invisibleFrames = 0
        self.frames_decoded = 0
        self.stats = []  # per frame: dict(encoding, size, groups, types histogram...) for reporting

    @staticmethod
    def _map(e):
        if e.status == -3:
            return UnsupportedOperationException(str(e))
        if e.status == -2:
            return InvalidBitstreamException(str(e))
        return e

    def getImageHeader(self):
        return self.info

    def _trace(self, stage, planes, fused):
        """test hook (tests/test_jvm_pin.py): `self.trace(frame_index, stage, planes, fused)` at the cut points where the pin-on-arrival
        harness makes the reference dump its planes; None (the default): nothing"""
        tr = getattr(self, "trace", None)
        if tr is not None:
            tr(self.frames_decoded - 1, stage, planes, fused)

    # -- frame-level pieces ---------------------------------------------------------------------------------------
    def _colors(self, fr):
        return 3 if (self.info.xyb_encoded or fr.encoding == VARDCT) else (1 if self.info.colour_space == CE_GRAY else 3)

    def _opsin(self):
        """OpsinInverseMatrix.getMatrix(bundle.prim, bundle.white) (JXLCodestreamDecoder.java:592-595)"""
        info = self.info
        conv = get_conversion_matrix(np.array(info.prim_xy, F), np.array(info.white_xy, F), PRI_SRGB, WP_D65)
        m = _mat_mul(conv, np.array(info.opsin_matrix, F).reshape(3, 3))
        bias = np.array(info.opsin_bias, F)
        cbrt = np.array([F(np.cbrt(np.float64(b))) for b in bias], F)
        return m.reshape(-1), bias, cbrt

    def _weights(self, fr):
        if fr.quant_all_default:
            return hfglobal.default_weights()
        params = []
        for i in range(17):
            q = self.fe.quant_params(i)
            if q["mode"] == 0:
                params.append(hfglobal.DEFAULT_PARAMS[i])
            else:
                params.append(dict(mode=q["mode"], denominator=q["denominator"],
                                   dct=None if q["dct"] is None else [list(r) for r in q["dct"]],
                                   par=None if q["par"] is None else [list(r) for r in q["par"]],
                                   p44=None if q["p44"] is None else [list(r) for r in q["p44"]]))
        return hfglobal.generate_weights(params)

    def _vardct_frame(self, fr, fuse_xyb, keep=None):
        p, weights, woffs, lfgroups, groups, hist = self._vardct_inputs(fr, fuse_xyb)
        self.stats[-1]["varblocks"] = {abi.TT_NAME[t]: int(n) for t, n in enumerate(hist) if n}
        if keep is not None:
            return self.backend.vardct(p, weights, woffs, lfgroups, groups(), keep=keep)
        planes = self.backend.vardct(p, weights, woffs, lfgroups, groups())
        return [np.ascontiguousarray(planes[c]) for c in range(3)]

    def _up_weights(self, k):
        info = self.info
        idx = {2: 0, 4: 1, 8: 2}[k]
        if info.custom_up[idx]:
            packed = self.fe.up_weights(idx)
        else:
            from .upweights import DEFAULT_UP
            packed = DEFAULT_UP[k]
        from . import host
        return host.getUpWeights(k, packed)

    def _chained_tail(self, fr, rp, buffers, colors, save, xyb_done):
        """Frame.upsample .. performColorTransforms (JXLCodestreamDecoder.java:628-637) of the three colour planes with the
        samples moving between host and device only where the next stage lives on the other side: upsampling, noise and the
        colour transforms are device stages on host.ResidentPlanes; the saveBeforeCT reference, patches and splines are host
        stages (as in the reference). `rp` is the VarDCT frame's resident result, or None when the colour planes start as the
        host arrays buffers[:3] (Modular frames). The extra channels in buffers[3:] are host arrays throughout."""
        info, be = self.info, self.backend
        moves = []

        def on_device():
            nonlocal rp
            if rp is None:  # every device stage works on float samples (Frame.java:221, :806; JXLCodestreamDecoder.java:262)
                rp = be.keep_planes(np.stack([self._to_float(buffers[c], info.bits_per_sample) for c in range(3)]))
                moves.append("h2d")
            return rp

        def on_host():
            nonlocal rp
            if rp is not None:
                planes = rp.download()
                for c in range(3):
                    buffers[c] = planes[c]
                rp = None
                moves.append("d2h")

        if fr.upsampling > 1:
            on_device().upsample(fr.upsampling, self._up_weights(fr.upsampling))
        if save and fr.save_before_ct:
            on_host()
            self.reference[fr.save_as_reference] = [b.copy() for b in buffers]
        if fr.num_patches:
            on_host()
            self._patches(fr, buffers, colors)
        if fr.has_splines:
            on_host()
            for c in range(3):
                buffers[c] = self._to_float(buffers[c], info.bits_per_sample).copy()
            render_splines(buffers, self.fe.splines(), fr.base_corr_x, fr.base_corr_b, buffers[0].shape[1], buffers[0].shape[0])
        if fr.has_noise:  # initializeNoise depends on the frame counters and the size only: its place before the patches is moot
            on_device().noise(fr.group_dim, (self.visibleFrames << 32) | self.invisibleFrames, np.array(fr.noise, F),
                              fr.base_corr_x, fr.base_corr_b)
        if info.xyb_encoded and not xyb_done:
            m, bias, cbrt = self._opsin()
            on_device().invertXYB(m, bias, cbrt, info.intensity_target)
        if fr.do_ycbcr:
            on_device().ycbcr()
        on_host()
        self.stats[-1]["plane_moves"] = moves

    def _vardct_inputs(self, fr, fuse_xyb):
        """the boundary tensors of one VarDCT frame: (jxl_vardct_params, weights, offsets, LF groups, group iterator)"""
        info, fe = self.info, self.fe
        p = abi.VarDCTParams()
        for c in range(3):
            p.jpeg_upsampling_y[c], p.jpeg_upsampling_x[c] = fr.jpeg_up_y[c], fr.jpeg_up_x[c]
        p.width, p.height = fr.padded_width, fr.padded_height
        stages = abi.STAGE_IDCT | abi.STAGE_GAB | abi.STAGE_EPF | (abi.STAGE_XYB if fuse_xyb else 0)
        p.stages = stages
        gs = F(F(65536.0) / F(fr.global_scale))  # HFCoefficients.java:270-275
        sf = [F(gs * F(math.pow(0.8, fr.xqm - 2.0))), gs, F(gs * F(math.pow(0.8, fr.bqm - 2.0)))]
        for c in range(3):
            p.scale_factor[c] = sf[c]
            p.quant_bias[c] = info.quant_bias[c]
            p.gab_w1[c], p.gab_w2[c] = fr.gab1[c], fr.gab2[c]
            p.epf_channel_scale[c] = fr.epf_channel_scale[c]
        p.quant_bias_numerator = info.quant_bias_numerator
        p.base_corr_x, p.base_corr_b, p.color_factor = fr.base_corr_x, fr.base_corr_b, fr.colour_factor
        p.gab, p.epf_iters = fr.gab, fr.epf_iters
        p.global_scale_f = gs
        for i in range(8):
            p.epf_sharp_lut[i] = fr.epf_sharp_lut[i]
        p.epf_pass0_sigma_scale, p.epf_pass2_sigma_scale = fr.epf_pass0_sigma, fr.epf_pass2_sigma
        p.epf_border_sad_mul = fr.epf_border_sad_mul
        m, bias, cbrt = self._opsin()
        p.xyb = 1 if fuse_xyb else 0
        for i in range(9):
            p.opsin_matrix[i] = m[i]
        for c in range(3):
            p.opsin_bias[c], p.cbrt_opsin_bias[c] = bias[c], cbrt[c]
        p.intensity_target = info.intensity_target
        p.transfer, p.out_format = abi.TRANSFER_NONE, abi.OUT_F32
        weights, woffs = self._weights(fr)
        adaptive = (fr.flags & (FLAG_SKIP_ADAPTIVE_LF | FLAG_USE_LF_FRAME)) == 0
        lfgroups, hist = [], np.zeros(27, np.int64)
        for i in range(fr.num_lf_groups):
            g = fe.lfgroup(i)
            g.update(lfg_y=i // fr.lf_group_cols, lfg_x=i % fr.lf_group_cols, lf=None,
                     scaled_dequant=list(fr.scaled_dequant), x_factor_lf=fr.x_factor_lf, b_factor_lf=fr.b_factor_lf,
                     adaptive_smoothing=adaptive)
            if fr.flags & FLAG_USE_LF_FRAME:
                g["lf"] = lf_from_lf_frame(self.lfBuffer[fr.lf_level], i // fr.lf_group_cols, i % fr.lf_group_cols, g["cells_h"],
                                           g["cells_w"], list(fr.jpeg_up_y), list(fr.jpeg_up_x), info.bits_per_sample)
                g["lf_quant"] = None
            g["block_yx"] = np.ascontiguousarray(g["block_yx"], np.int32)
            sel = g["dct_select"][g["block_yx"][:, 0], g["block_yx"][:, 1]]
            hist += np.bincount(sel, minlength=27)[:27]
            lfgroups.append(g)

        def groups():
            for pass_ in range(fr.num_passes):
                for grp in range(fr.num_groups):
                    yield pass_, grp, fe.coeffs(pass_, grp)
        return p, weights, woffs, lfgroups, groups, hist

    def _modular_buffers(self, fr, buffers, colors):
        """modular channels -> frame buffers (Frame.decodeFrame :430-455)"""
        info, fe = self.info, self.fe
        n_mod = fr.num_modular_channels
        chans = [fe.modular_channel(i)[0] for i in range(n_mod)]
        self._trace("mod", chans, False)  # ModularStream.getDecodedBuffer after applyTransforms (Frame.java:427-428)
        c_map = (1, 0, 2)
        for c in range(n_mod):
            is_mod_color = fr.encoding == MODULAR and c < colors
            is_mod_xyb = bool(info.xyb_encoded) and is_mod_color
            c_out = (c_map[c] if is_mod_xyb else c) + len(buffers) - n_mod
            h, w = fr.height, fr.width
            src = chans[c][:h, :w]
            scale = fr.lf_dequant[c_out] if is_mod_xyb else 1.0
            if is_mod_xyb and c == 2:
                val = self.backend.modular_to_float(np.ascontiguousarray(chans[0][:h, :w]), np.ascontiguousarray(src), scale)
            elif buffers[c_out].dtype == np.float32:
                val = self.backend.modular_to_float(np.ascontiguousarray(src), None, scale)
            else:
                val = src
            buffers[c_out][:h, :w] = val

    def _blend_frame(self, fr, frame_buffers, colors_frame):
        """JXLCodestreamDecoder.blendFrame + blendBuffers (:424-537)"""
        info = self.info
        ih, iw = info.height, info.width
        colors = 1 if info.colour_space == CE_GRAY else 3
        py, px = min(max(fr.y0, 0), ih), min(max(fr.x0, 0), iw)       # Point.inBounds
        fy, fx = py - fr.y0, px - fr.x0
        ly, lx = fr.y0 + fr.height * fr.upsampling, fr.x0 + fr.width * fr.upsampling  # bounds after Frame.upsample()
        bh, bw = min(ly, ih) - py, min(lx, iw) - px
        if bh <= 0 or bw <= 0:
            return
        has_extra = info.num_extra > 0
        for c in range(len(self.canvas)):
            if c >= colors:
                e = c - colors
                mode, alpha_ch, clamp, source = fr.ec_blend_mode[e], fr.ec_blend_alpha[e], fr.ec_blend_clamp[e], fr.ec_blend_source[e]
            else:
                mode, alpha_ch, clamp, source = fr.blend_mode, fr.blend_alpha, fr.blend_clamp, fr.blend_source
            ref_buffers = self.reference[source]
            self._blend_buffers(c, frame_buffers, ref_buffers, (py, px), (fy, fx), (py, px), (bh, bw), colors_frame,
                                mode, alpha_ch, bool(clamp), patch=False, canvas_list=self.canvas)

    def _depth_of(self, idx):
        colors = 1 if self.info.colour_space == CE_GRAY else 3
        return self.info.bits_per_sample if idx < colors else self.info.ec_bits[idx - colors]

    @staticmethod
    def _to_float(a, depth):
        if a.dtype == np.float32:
            return a
        return (a.astype(F) * F(F(1) / F((1 << depth) - 1))).astype(F)

    def _blend_buffers(self, idx, frame_buffers, ref_buffers, patch_start, frame_off, ref_off, size, frame_colors, mode,
                       alpha_channel, clamp, patch, canvas_list):
        info = self.info
        colors = 1 if info.colour_space == CE_GRAY else 3
        fb_idx = (1 if idx == 0 else idx + 2) if colors != frame_colors else idx
        ex = idx - colors
        is_extra = ex >= 0
        has_extra = info.num_extra > 0
        is_alpha = is_extra and info.ec_type[ex] == 0
        premult = has_extra and bool(info.ec_alpha_associated[alpha_channel])
        depth = self._depth_of(idx)
        canvas = canvas_list[idx]
        frame_buffer = frame_buffers[fb_idx]
        if canvas.dtype != frame_buffer.dtype:
            frame_buffer = frame_buffers[fb_idx] = self._to_float(frame_buffer, depth)
            canvas = canvas_list[idx] = self._to_float(canvas, depth)
        rect = (size[0], size[1], patch_start[0], patch_start[1], frame_off[0], frame_off[1], ref_off[0], ref_off[1])
        if mode == abi.BLEND_REPLACE and not patch or (ref_buffers is None and mode == abi.BLEND_ADD and not patch):
            canvas_list[idx] = self.backend.blend(abi.BLEND_REPLACE, canvas, frame_buffer, None, rect)
            return
        if ref_buffers is None:
            if patch:
                return
            ref_buffers = [None] * len(canvas_list)
        if ref_buffers[idx] is None:
            ref_buffers[idx] = np.zeros(canvas.shape, canvas.dtype)
        ref_alpha = frame_alpha = None
        a_idx_ref, a_idx_frame = colors + alpha_channel, frame_colors + alpha_channel
        pmode = mode
        if patch:  # Patch blend modes 0..7 -> frame blend modes (JXLCodestreamDecoder.java:478-494)
            if mode == 0:
                return
            pmode, below = {5: (abi.BLEND_BLEND, True), 6: (abi.BLEND_MULADD, False), 7: (abi.BLEND_MULADD, True)}.get(mode, (mode - 1, False))
        else:
            below = False
        if has_extra and mode in (abi.BLEND_BLEND, abi.BLEND_MULADD):
            a_depth = info.ec_bits[alpha_channel]
            if mode == abi.BLEND_BLEND:
                if ref_buffers[a_idx_ref] is None:
                    ref_buffers[a_idx_ref] = np.zeros(canvas.shape, F)
                ref_buffers[a_idx_ref] = self._to_float(ref_buffers[a_idx_ref], a_depth)
            frame_buffers[a_idx_frame] = self._to_float(frame_buffers[a_idx_frame], a_depth)
        if has_extra:
            ref_alpha = ref_buffers[a_idx_ref]
            frame_alpha = frame_buffers[a_idx_frame]
        should_cast = mode == abi.BLEND_MULT or (mode == abi.BLEND_BLEND and has_extra) or \
            (mode == abi.BLEND_MULADD and has_extra and not is_alpha)
        ref_buffer = ref_buffers[idx]
        if should_cast or ref_buffer.dtype != frame_buffer.dtype:
            frame_buffer = frame_buffers[fb_idx] = self._to_float(frame_buffer, depth)
            canvas = canvas_list[idx] = self._to_float(canvas, depth)
            ref_buffer = ref_buffers[idx] = self._to_float(ref_buffer, depth)
        old_buffer, new_buffer = (ref_buffer, frame_buffer) if below else (frame_buffer, ref_buffer)
        fa = frame_alpha if (frame_alpha is not None and frame_alpha.dtype == np.float32) else None
        ra = ref_alpha if (ref_alpha is not None and ref_alpha.dtype == np.float32) else None
        canvas_list[idx] = self.backend.blend(pmode, canvas, old_buffer, new_buffer, rect, frameAlpha=fa, refAlpha=ra,
                                              isAlpha=is_alpha, hasExtra=has_extra, clamp=clamp, premult=premult)

    def _patches(self, fr, frame_buffers, frame_colors):
        """JXLCodestreamDecoder.computePatches (:204-254)"""
        info = self.info
        colors = 1 if info.colour_space == CE_GRAY else 3
        for i in range(fr.num_patches):
            p = self.fe.patch(i)
            if p["ref"] > 3:
                raise InvalidBitstreamException("Patch out of range")
            ref = self.reference[p["ref"]]
            if ref is None:
                continue
            if p["y0"] + p["h"] > ref[0].shape[0] or p["x0"] + p["w"] > ref[0].shape[1]:
                raise InvalidBitstreamException("Patch too large")
            for j in range(p["positions"].shape[0]):
                y0, x0 = int(p["positions"][j, 0]), int(p["positions"][j, 1])
                if y0 < 0 or x0 < 0 or p["h"] + y0 > frame_buffers[0].shape[0] or p["w"] + x0 > frame_buffers[0].shape[1]:
                    raise InvalidBitstreamException("Patch size out of bounds")
                for d in range(colors + info.num_extra):
                    c = 0 if d < colors else d - colors + 1
                    mode, alpha, clamp = (int(v) for v in p["blend"][j, c])
                    if mode == 0:
                        continue
                    self._blend_buffers(d, frame_buffers, ref, (y0, x0), (y0, x0), (p["y0"], p["x0"]), (p["h"], p["w"]),
                                        frame_colors, mode, alpha, bool(clamp), patch=True, canvas_list=frame_buffers)

    # -- the decode loop (JXLCodestreamDecoder.decode :546-677) ----------------------------------------------------
    def decode(self):
        info, be = self.info, self.backend
        colors_img = 1 if info.colour_space == CE_GRAY else 3
        if self.canvas is None:
            self.canvas = [None] * (colors_img + info.num_extra)
        produced = False
        while True:
            try:
                fr = self.fe.next_frame(be.squeeze, be.rct)
            except frontend.FrontendError as e:
                raise self._map(e)
            if fr is None:
                break
            produced = True
            self.frames_decoded += 1
            self.stats.append(dict(encoding="vardct" if fr.encoding == VARDCT else "modular", width=fr.width, height=fr.height,
                                   groups=fr.num_groups, passes=fr.num_passes))
            if fr.flags & FLAG_USE_LF_FRAME and self.lfBuffer[fr.lf_level] is None:
                raise InvalidBitstreamException("LF Level too large")  # JXLCodestreamDecoder.java:613-614
            colors = self._colors(fr)
            simple = fr.upsampling == 1 and not fr.num_patches and not fr.has_splines and not fr.has_noise and \
                not (fr.save_before_ct and not fr.is_last)
            ph, pw = fr.padded_height, fr.padded_width
            buffers = []
            for c in range(colors + info.num_extra):
                if c < colors:
                    is_float = bool(info.xyb_encoded) or fr.encoding == VARDCT or info.exp_bits != 0
                else:
                    is_float = info.ec_exp_bits[c - colors] != 0
                buffers.append(np.zeros((ph, pw), F if is_float else np.int32))
            xyb_done = False
            rp = None
            resident = getattr(be, "resident", False) and colors == 3  # row f4: the stages after decodeFrame chained on the device
            if fr.encoding == VARDCT:
                # an LF frame's buffers are read back as XYB LF coefficients (LFCoefficients.java:44-57) and are stored
                # BEFORE performColorTransforms (JXLCodestreamDecoder.java:615-617): never fuse the inverse XYB into them
                fuse_xyb = bool(info.xyb_encoded) and simple and fr.lf_level == 0 and fr.type != LF_FRAME
                # frames with stages between decodeFrame and the colour transform keep their colour planes on the device
                # through those stages (row f4); LF frames / lfBuffer consumers need the padded planes on the host
                if not simple and resident and fr.lf_level == 0 and fr.type != LF_FRAME:
                    rp = self._vardct_frame(fr, False, keep=(fr.height, fr.width))
                    for c in range(3):
                        buffers[c] = np.zeros((fr.height, fr.width), F)  # stand-ins until the chained tail downloads
                else:
                    planes = self._vardct_frame(fr, fuse_xyb)
                    xyb_done = fuse_xyb
                    for c in range(3):
                        buffers[c] = planes[c]
            self._modular_buffers(fr, buffers, colors)
            # (trace: the cut points of integration/jvm_pin/StageDump.java. For a VarDCT frame the colour planes in `buffers` are
            # already the fused kernel's result -- `fused` tells the listener to take the stages of planes 0..2 elsewhere)
            fused = fr.encoding == VARDCT
            self._trace("idct", buffers, fused)
            self._trace("sub", buffers, fused)  # Frame.invertSubsampling: VarDCT colour planes only (inside the backend call)

            def colour_planes_to_float():
                # Frame.performGabConvolution / performEdgePreservingFilter cast integer colour planes to float first
                # (Frame.java:519: ImageBuffer.castToFloat = v * (1f / maxValue)); one-colour frames keep one plane (the backends
                # feed the three-channel kernels three copies: Frame.java:642,661 read channel 0 in all three rounds)
                for c in range(colors):
                    if buffers[c].dtype != np.float32:
                        maxv = (1 << info.bits_per_sample) - 1
                        buffers[c] = be.modular_to_float(np.ascontiguousarray(buffers[c], np.int32), None, float(F(1) / F(maxv)))
                return np.stack(buffers[:colors])
            if fr.encoding == MODULAR and fr.gab:
                planes = be.gab(colour_planes_to_float(), list(fr.gab1), list(fr.gab2))
                for c in range(colors):
                    buffers[c] = np.ascontiguousarray(planes[c])
            self._trace("gab", buffers, fused)
            if fr.encoding == MODULAR and fr.epf_iters > 0:
                sigma = F(F(1) / F(fr.epf_sigma_modular))  # Frame.java:573-575
                planes = be.epf(colour_planes_to_float(), fr.epf_iters, None, float(sigma),
                                dict(channel_scale=list(fr.epf_channel_scale), pass0=fr.epf_pass0_sigma,
                                     pass2=fr.epf_pass2_sigma, border_sad_mul=fr.epf_border_sad_mul))
                for c in range(colors):
                    buffers[c] = np.ascontiguousarray(planes[c])
            self._trace("epf", buffers, fused)
            if fr.lf_level > 0:  # JXLCodestreamDecoder.java:616-617: the frame's buffers as they stand after decodeFrame
                self.lfBuffer[fr.lf_level - 1] = [np.array(b, copy=True) for b in buffers]
            # crop to the frame bounds: everything after the restoration filters works on header.bounds
            buffers = [np.ascontiguousarray(b[:fr.height, :fr.width]) for b in buffers]
            if fr.type == LF_FRAME:
                continue
            save = (fr.save_as_reference != 0 or fr.duration == 0) and not fr.is_last and fr.type != LF_FRAME
            visible = fr.type in (REGULAR_FRAME, SKIP_PROGRESSIVE) and (fr.duration != 0 or fr.is_last)
            if visible:
                self.visibleFrames += 1
                self.invisibleFrames = 0
            else:
                self.invisibleFrames += 1
            # Frame.upsample
            for c in range(len(buffers)):
                k = fr.upsampling if c < colors else fr.ec_upsampling[c - colors]
                if k > 1 and not (resident and c < 3):
                    wts = self._up_weights(k)
                    depth = info.bits_per_sample if c < colors else info.ec_bits[c - colors]
                    buffers[c] = be.upsample(self._to_float(buffers[c], depth), k, wts)
            noise = None
            if resident:
                self._chained_tail(fr, rp, buffers, colors, save, xyb_done)
            elif fr.has_noise:
                h, w = buffers[0].shape
                noise = be.noise_init(h, w, (self.visibleFrames << 32) | self.invisibleFrames, fr.group_dim, colors)
            if not resident and save and fr.save_before_ct:
                self.reference[fr.save_as_reference] = [b.copy() for b in buffers]
            if not resident:
                self._patches(fr, buffers, colors)
            if not resident and fr.has_splines:  # Frame.renderSplines (host-side, as in the reference)
                for c in range(3):
                    buffers[c] = self._to_float(buffers[c], info.bits_per_sample).copy()
                render_splines(buffers, self.fe.splines(), fr.base_corr_x, fr.base_corr_b, buffers[0].shape[1], buffers[0].shape[0])
            if noise is not None:
                planes = np.stack([self._to_float(buffers[c], info.bits_per_sample) for c in range(3)])
                planes = be.noise_add(planes, noise, np.array(fr.noise, F), fr.base_corr_x, fr.base_corr_b)
                for c in range(3):
                    buffers[c] = np.ascontiguousarray(planes[c])
            # performColorTransforms
            if not resident and ((info.xyb_encoded and not xyb_done) or fr.do_ycbcr):
                planes = np.stack([self._to_float(buffers[c], info.bits_per_sample) for c in range(3)])
                if info.xyb_encoded and not xyb_done:
                    m, bias, cbrt = self._opsin()
                    planes = be.xyb(planes, m, bias, cbrt, info.intensity_target)
                if fr.do_ycbcr:
                    planes = be.ycbcr(planes)
                for c in range(3):
                    buffers[c] = np.ascontiguousarray(planes[c])
            self._trace("xyb", buffers, False)  # JXLCodestreamDecoder.java:637: the frame's buffers after performColorTransforms
            if self.canvas[0] is None:
                for c in range(len(self.canvas)):
                    self.canvas[c] = np.zeros((info.height, info.width), buffers[0].dtype)
            if fr.type in (REGULAR_FRAME, SKIP_PROGRESSIVE):
                if any(self.reference[i] is self.canvas and i != fr.save_as_reference for i in range(4)):
                    self.canvas = [b.copy() for b in self.canvas]
                self._blend_frame(fr, buffers, colors)
            if save and not fr.save_before_ct:
                self.reference[fr.save_as_reference] = self.canvas
            if fr.is_last or fr.duration != 0:
                break
        if not produced:
            return None
        oriented = [be.orient(np.ascontiguousarray(b), info.orientation) if info.orientation != 1 else b for b in self.canvas]
        return JXLImage(oriented, info, be)


# ---- PNGWriter (J/io/PNGWriter.java) ---------------------------------------------------------------------------
class PNGWriter:
    def __init__(self, image, bitDepth=-1, hdr=False, peakDetect=PEAK_DETECT_AUTO, deflateLevel=6):
        if bitDepth <= 0:
            bitDepth = 16 if (hdr or image.info.bits_per_sample > 8) else 8
        if bitDepth not in (8, 16):
            raise ValueError("PNG only supports 8 and 16")
        self.hdr = hdr
        gray = image.colorEncoding == CE_GRAY
        primaries = PRI_BT2100 if hdr else PRI_SRGB
        tf = TF_PQ if hdr else TF_SRGB
        self.has_icc = image.has_icc
        if not image.has_icc:
            image = image.transform(primaries, WP_D65, tf, peakDetect)
        self.bitDepth = bitDepth
        self.width, self.height = image.getWidth(), image.getHeight()
        self.alphaIndex = image.getAlphaIndex()
        self.colorChannels = 1 if gray else 3
        self.colorMode = (4 if self.alphaIndex >= 0 else 0) if gray else (6 if self.alphaIndex >= 0 else 2)
        self.deflateLevel = deflateLevel
        planes = image.getBuffer(False)
        color = [np.ascontiguousarray(planes[c]) for c in range(self.colorChannels)]
        alpha = np.ascontiguousarray(planes[self.colorChannels + self.alphaIndex]) if self.alphaIndex >= 0 else None
        tagged = [image.getTaggedBitDepth(c) for c in range(self.colorChannels)]
        if alpha is not None:
            tagged.append(image.getTaggedBitDepth(self.colorChannels + self.alphaIndex))
        tagged += [bitDepth] * (4 - len(tagged))
        # PNGWriter.java:79-111 + the writeIDAT sample order: one device pass
        self.samples = image.backend.pack(color, bitDepth, alpha, image.isAlphaPremultiplied() and alpha is not None, tagged, True)

    @staticmethod
    def _chunk(tag, payload):
        body = tag + payload
        return struct.pack(">I", len(payload)) + body + struct.pack(">I", zlib.crc32(body) & 0xffffffff)

    def write(self, out):
        out.write(b"\x89PNG\r\n\x1a\n")
        out.write(self._chunk(b"IHDR", struct.pack(">IIBBBBB", self.width, self.height, self.bitDepth, self.colorMode, 0, 0, 0)))
        if not self.has_icc and not self.hdr:
            out.write(self._chunk(b"sRGB", b"\x01"))
        if self.hdr:
            out.write(self._chunk(b"cICP", bytes([9, 16, 0, 1])))  # BT.2100 PQ full range (the reference embeds an ICC profile)
        rows = self.samples.reshape(self.height, -1).view(np.uint8)
        raw = np.concatenate([np.zeros((self.height, 1), np.uint8), rows], axis=1).tobytes()  # filter type 0 per row
        out.write(self._chunk(b"IDAT", zlib.compress(raw, self.deflateLevel)))
        out.write(self._chunk(b"IEND", b""))


def load_vardct_frame(source, ctx, transfer=abi.TRANSFER_NONE, out_format=abi.OUT_F32):
    """Parse the first frame of a VarDCT .jxl file with the front-end and stage it in a host.Frame on `ctx` (inputs
    resident, nothing run yet): the real-bitstream workload of bench.py. Returns (host.Frame, stats dict)."""
    from . import host
    if isinstance(source, (bytes, bytearray)):
        data = bytes(source)
    else:
        with open(source, "rb") as f:
            data = f.read()
    dec = JXLDecoder.__new__(JXLDecoder)
    dec.fe = frontend.Frontend(data)
    dec.info = dec.fe.image
    dec.backend = None
    fr = dec.fe.next_frame(None, None)
    if fr is None or fr.encoding != VARDCT:
        raise ValueError("first frame is not a VarDCT frame")
    p, weights, woffs, lfgroups, groups, hist = dec._vardct_inputs(fr, bool(dec.info.xyb_encoded))
    if transfer != abi.TRANSFER_NONE or out_format != abi.OUT_F32:
        p.stages |= abi.STAGE_OUT
        p.transfer, p.out_format = transfer, out_format
    hf = host.Frame(ctx, p, weights, woffs)
    for g in lfgroups:
        hf.setLFGroup(g)
        if g.get("lf_quant") is not None:
            hf.setLFGroupQuant(g["lfg_y"], g["lfg_x"], g["lf_quant"], g["scaled_dequant"],
                               g["extra_precision"], g["x_factor_lf"], g["b_factor_lf"], g["adaptive_smoothing"])
    for pass_, grp, q in groups():
        hf.putGroup(pass_, grp, q)
    stats = dict(width=fr.width, height=fr.height, padded_width=fr.padded_width, padded_height=fr.padded_height, groups=fr.num_groups,
                 passes=fr.num_passes, epf_iters=fr.epf_iters, gab=fr.gab,
                 varblocks={abi.TT_NAME[t]: int(n) for t, n in enumerate(hist) if n})
    return hf, stats

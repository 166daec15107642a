"""Host-side producer of the HF quantisation weight tables (the `HFGlobal.weights` input of
the boundary). On the reference this is J/frame/vardct/HFGlobal.java:42-77 (band
interpolation), :79-188 (the default parameter sets of the JPEG XL spec) and :304-432
(per-mode weight layout and the final reciprocal). It runs once per frame on the host
and is NOT part of the device hot path; it exists here so that synthetic frames carry
realistic weights without a JVM. float32 arithmetic mirrors the Java expression order;
`Math.pow` is evaluated in double with libm (Java specifies it only to 1 ulp, so the
weights are boundary *inputs*, identical for oracle and device).
"""
import math

import numpy as np

from . import abi

F = np.float32

MODE_HORNUSS, MODE_DCT2, MODE_DCT4, MODE_DCT4_8, MODE_AFV, MODE_DCT, MODE_RAW = 1, 2, 3, 4, 5, 6, 7

_SEQ_A = [-1.025, -0.78, -0.65012, -0.19041574084286472, -0.20819395464, -0.421064, -0.32733845535848671]
_SEQ_B = [-0.3041958212306401, -0.3633036457487539, -0.35660379990111464, -0.3443074455424403,
          -0.33699592683512467, -0.30180866526242109, -0.27321683125358037]
_SEQ_C = [-1.2, -1.2, -0.8, -0.7, -0.7, -0.4, -0.5]
_DCT4X4 = [[2200.0, 0.0, 0.0, 0.0], [392.0, 0.0, 0.0, 0.0], [112.0, -0.25, -0.25, -0.5]]
_DCT4X8 = [[2198.050556016380522, -0.96269623020744692, -0.76194253026666783, -0.6551140670773547],
           [764.3655248643528689, -0.92630200888366945, -0.9675229603596517, -0.27845290869168118],
           [527.107573587542228, -1.4594385811273854, -1.450082094097871593, -1.5843722511996204]]


def _big(a, b, c):
    return [[a] + _SEQ_A, [b] + _SEQ_B, [c] + _SEQ_C]


# index = parameterIndex; dict(mode, dct=dctParam, par=param, p44=params4x4)   (HFGlobal.java:79-188)
DEFAULT_PARAMS = [
    dict(mode=MODE_DCT, dct=[[3150.0, 0.0, -0.4, -0.4, -0.4, -2.0], [560.0, 0.0, -0.3, -0.3, -0.3, -0.3],
                             [512.0, -2.0, -1.0, 0.0, -1.0, -2.0]]),
    dict(mode=MODE_HORNUSS, par=[[280.0, 3160.0, 3160.0], [60.0, 864.0, 864.0], [18.0, 200.0, 200.0]]),
    dict(mode=MODE_DCT2, par=[[3840.0, 2560.0, 1280.0, 640.0, 480.0, 300.0], [960.0, 640.0, 320.0, 180.0, 140.0, 120.0],
                              [640.0, 320.0, 128.0, 64.0, 32.0, 16.0]]),
    dict(mode=MODE_DCT4, dct=_DCT4X4, par=[[1.0, 1.0]] * 3, p44=_DCT4X4),
    dict(mode=MODE_DCT, dct=[
        [8996.8725711814115328, -1.3000777393353804, -0.49424529824571225, -0.439093774457103443,
         -0.6350101832695744, -0.90177264050827612, -1.6162099239887414],
        [3191.48366296844234752, -0.67424582104194355, -0.80745813428471001, -0.44925837484843441,
         -0.35865440981033403, -0.31322389111877305, -0.37615025315725483],
        [1157.50408145487200256, -2.0531423165804414, -1.4, -0.50687130033378396, -0.42708730624733904,
         -1.4856834539296244, -4.9209142884401604]]),
    dict(mode=MODE_DCT, dct=[
        [15718.40830982518931456, -1.025, -0.98, -0.9012, -0.4, -0.48819395464, -0.421064, -0.27],
        [7305.7636810695983104, -0.8041958212306401, -0.7633036457487539, -0.55660379990111464,
         -0.49785304658857626, -0.43699592683512467, -0.40180866526242109, -0.27321683125358037],
        [3803.53173721215041536, -3.060733579805728, -2.0413270132490346, -2.0235650159727417,
         -0.5495389509954993, -0.4, -0.4, -0.3]]),
    dict(mode=MODE_DCT, dct=[[7240.7734393502, -0.7, -0.7, -0.2, -0.2, -0.2, -0.5],
                             [1448.15468787004, -0.5, -0.5, -0.5, -0.2, -0.2, -0.2],
                             [506.854140754517, -1.4, -0.2, -0.5, -0.5, -1.5, -3.6]]),
    dict(mode=MODE_DCT, dct=[
        [16283.2494710648897, -1.7812845336559429, -1.6309059012653515, -1.0382179034313539, -0.85, -0.7, -0.9,
         -1.2360638576849587],
        [5089.15750884921511936, -0.320049391452786891, -0.35362849922161446, -0.30340000000000003, -0.61, -0.5, -0.5,
         -0.6],
        [3397.77603275308720128, -0.321327362693153371, -0.34507619223117997, -0.70340000000000003, -0.9, -1.0, -1.0,
         -1.1754605576265209]]),
    dict(mode=MODE_DCT, dct=[
        [13844.97076442300573, -0.97113799999999995, -0.658, -0.42026, -0.22712, -0.2206, -0.226, -0.6],
        [4798.964084220744293, -0.61125308982767057, -0.83770786552491361, -0.79014862079498627,
         -0.2692727459704829, -0.38272769465388551, -0.22924222653091453, -0.20719098826199578],
        [1807.236946760964614, -1.2, -1.2, -0.7, -0.7, -0.7, -0.4, -0.5]]),
    dict(mode=MODE_DCT4_8, dct=_DCT4X8, par=[[1.0], [1.0], [1.0]]),
    dict(mode=MODE_AFV, dct=_DCT4X8, p44=_DCT4X4,
         par=[[3072.0, 3072.0, 256.0, 256.0, 256.0, 414.0, 0.0, 0.0, 0.0],
              [1024.0, 1024.0, 50.0, 50.0, 50.0, 58.0, 0.0, 0.0, 0.0],
              [384.0, 384.0, 12.0, 12.0, 12.0, 22.0, -0.25, -0.25, -0.25]]),
    dict(mode=MODE_DCT, dct=_big(23966.1665298448605, 8380.19148390090414, 4493.02378009847706)),
    dict(mode=MODE_DCT, dct=_big(15358.89804933239925, 5597.360516150652990, 2919.961618960011210)),
    dict(mode=MODE_DCT, dct=_big(47932.3330596897210, 16760.38296780180828, 8986.04756019695412)),
    dict(mode=MODE_DCT, dct=_big(30717.796098664792, 11194.72103230130598, 5839.92323792002242)),
    dict(mode=MODE_DCT, dct=_big(95864.6661193794420, 33520.76593560361656, 17972.09512039390824)),
    dict(mode=MODE_DCT, dct=_big(61435.5921973295970, 24209.44206460261196, 12979.84647584004484)),
]

_AFV_FREQS = [0, 0, 0.8517778890324296, 5.37778436506804, 0, 0, 4.734747904497923, 5.449245381693219,
              1.6598270267479331, 4, 7.275749096817861, 10.423227632456525, 2.662932286148962, 7.630657783650829,
              8.962388608184032, 12.97166202570235]


def _quant_mult(v):
    v = F(v)
    return F(1.0) + v if v >= 0 else F(1.0) / (F(1.0) - v)


def _interpolate(scaled_pos, bands):
    ln = len(bands) - 1
    if ln == 0:
        return bands[0]
    si = int(scaled_pos)
    frac = F(scaled_pos) - F(si)
    if si + 1 > ln:
        return bands[ln]
    a, b = bands[si], bands[si + 1]
    return F(a * F(math.pow(float(F(b / a)), float(frac))))


def dct_quant_weights(height, width, params):
    """HFGlobal.getDCTQuantWeights (HFGlobal.java:59-77), vectorised over (y, x); every
    float32 operation is the same IEEE operation as the scalar Java expression."""
    bands = [F(params[0])]
    for i in range(1, len(params)):
        bands.append(F(bands[i - 1] * _quant_mult(params[i])))
    sqrt2 = F(math.sqrt(2.0))
    scale = F(F(len(bands) - 1) / F(sqrt2 + F(1e-6)))
    dy = (np.arange(height, dtype=F) * scale) / F(height - 1)
    dx = (np.arange(width, dtype=F) * scale) / F(width - 1)
    d2 = (dx * dx)[None, :] + (dy * dy)[:, None]
    dist = np.sqrt(d2.astype(np.float64)).astype(F)
    ln = len(bands) - 1
    if ln == 0:
        return np.full((height, width), bands[0], F)
    si = dist.astype(np.int64)
    frac = dist - si.astype(F)
    hi = si + 1 > ln
    sic = np.minimum(si, ln - 1)
    barr = np.array(bands, F)
    a = barr[sic]
    ratio = (barr[sic + 1] / a).astype(F)
    pw = np.array([math.pow(r, f) for r, f in zip(ratio.ravel().tolist(), frac.ravel().tolist())],
                  np.float64).reshape(dist.shape).astype(F)
    return np.where(hi, barr[ln], (a * pw).astype(F)).astype(F)


def _afv_weights(prm, c):
    """HFGlobal.getAFVTransformWeights (HFGlobal.java:304-345)"""
    w48 = dct_quant_weights(4, 8, prm["dct"][c])
    w44 = dct_quant_weights(4, 4, prm["p44"][c])
    low, high = F(0.8517778890324296), F(12.97166202570235)
    par = [F(v) for v in prm["par"][c]]
    bands = [par[5]]
    for i in range(1, 4):
        bands.append(F(bands[i - 1] * _quant_mult(par[i + 5])))
    w = np.zeros((8, 8), F)
    w[0, 0] = 1.0
    w[1, 0], w[0, 1], w[2, 0], w[0, 2], w[2, 2] = par[0], par[1], par[2], par[3], par[4]
    for y in range(4):
        for x in range(4):
            if x < 2 and y < 2:
                continue
            pos = F(F(F(_AFV_FREQS[y * 4 + x]) - low) / F(high - low))
            w[2 * x, 2 * y] = _interpolate(pos, bands)
        for x in range(8):
            if x == 0 and y == 0:
                continue
            w[2 * y + 1, x] = w48[y, x]
        for x in range(4):
            if x == 0 and y == 0:
                continue
            w[2 * y, 2 * x + 1] = w44[y, x]
    return w


def generate_weights(params=None):
    """HFGlobal.generateWeights for all 17 parameter sets (HFGlobal.java:347-432).

    Returns (flat float32 array of reciprocal weights, int32[51] offsets) in the layout of
    jxl_vardct_set_weights: set p, channel c at offs[p*3+c], matrixHeight x matrixWidth row-major.
    """
    params = params or DEFAULT_PARAMS
    chunks, offs, pos = [], np.zeros(51, np.int32), 0
    for idx in range(17):
        prm = params[idx]
        mh, mw = abi.tt_matrix_size(idx)
        for c in range(3):
            mode = prm["mode"]
            if mode == MODE_DCT:
                w = dct_quant_weights(mh, mw, prm["dct"][c])
            elif mode == MODE_DCT4:
                w4 = dct_quant_weights(4, 4, prm["dct"][c])
                w = np.zeros((8, 8), F)
                for y in range(8):
                    for x in range(8):
                        w[y, x] = w4[y // 2, x // 2]
                w[1, 0] = F(w[1, 0] / F(prm["par"][c][0]))
                w[0, 1] = F(w[0, 1] / F(prm["par"][c][0]))
                w[1, 1] = F(w[1, 1] / F(prm["par"][c][1]))
            elif mode == MODE_DCT2:
                par = [F(v) for v in prm["par"][c]]
                w = np.zeros((8, 8), F)
                w[0, 0] = 1.0
                w[0, 1] = w[1, 0] = par[0]
                w[1, 1] = par[1]
                for y in range(2):
                    for x in range(2):
                        w[y, x + 2] = w[x + 2, y] = par[2]
                        w[y + 2, x + 2] = par[3]
                for y in range(4):
                    for x in range(4):
                        w[y, x + 4] = w[x + 4, y] = par[4]
                        w[y + 4, x + 4] = par[5]
            elif mode == MODE_HORNUSS:
                par = [F(v) for v in prm["par"][c]]
                w = np.full((8, 8), par[0], F)
                w[1, 1] = par[2]
                w[0, 1] = w[1, 0] = par[1]
                w[0, 0] = 1.0
            elif mode == MODE_DCT4_8:
                w48 = dct_quant_weights(4, 8, prm["dct"][c])
                w = np.zeros((8, 8), F)
                for y in range(8):
                    for x in range(8):
                        w[y, x] = w48[y // 2, x]
                w[1, 0] = F(w[1, 0] / F(prm["par"][c][0]))
            elif mode == MODE_AFV:
                w = _afv_weights(prm, c)
            elif mode == MODE_RAW:  # HFGlobal.java:407-416: the table itself times the denominator, no reciprocal
                w = (np.array(prm["par"][c], F).reshape(mh, mw) * F(prm.get("denominator", 1.0))).astype(F)
            else:
                raise ValueError("unsupported quant-weight mode %r" % mode)
            assert w.shape == (mh, mw)
            if mode != MODE_RAW:
                if not (np.all(w > 0) and np.all(np.isfinite(w))):
                    raise ValueError("Negative or infinite weight: %d, %d" % (idx, c))  # HFGlobal.java:425-426
                w = (F(1.0) / w).astype(F)
            offs[idx * 3 + c] = pos
            chunks.append(w.ravel())
            pos += w.size
    return np.concatenate(chunks).astype(F), offs


_cache = None


def default_weights():
    global _cache
    if _cache is None:
        _cache = generate_weights()
    return _cache

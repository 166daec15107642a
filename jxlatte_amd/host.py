"""Host-side mirror of the reference's interface for the transform stage, on top of the C-ABI.

Names, argument meaning and error behaviour follow the Java classes the path lives in
(J/ = java/com/traneptora/jxlatte/): `MathHelper.inverseDCT2D`, `Frame.performGabConvolution`,
`Frame.performEdgePreservingFilter`, `OpsinInverseMatrix.invertXYB`, `PassGroup.invertVarDCT`
(through `Frame.decodePassGroups`), `ModularStream.applyTransforms`,
`ModularChannel.inverse{Horizontal,Vertical}Squeeze`, `JXLImage.transfer`. numpy planes stand in for
the Java `float[][]` / `int[][]` row arrays. Every call runs on the GPU through libjxlatte_amd.so;
nothing here computes pixels on the CPU.
"""
import ctypes as C

import numpy as np

from . import abi
from ._lib import Context, check  # noqa: F401  (re-export)


def _p3(planes, ctype):
    arr = (C.POINTER(ctype) * 3)()
    for c in range(3):
        arr[c] = abi.ptr(planes[c], ctype)
    return arr


def _planes(a, dtype):
    a = np.ascontiguousarray(a, dtype)
    if a.ndim != 3 or a.shape[0] != 3:
        raise ValueError("expected planes of shape (3, H, W)")
    return a


class MathHelper:
    """J/util/MathHelper.java"""

    @staticmethod
    def inverseDCT2D(ctx, src, transposed=False):
        src = np.ascontiguousarray(src, np.float32)
        h, w = src.shape
        dst = np.empty((w, h) if transposed else (h, w), np.float32)
        ctx.call("jxl_stage_idct2d", abi.fptr(src), abi.fptr(dst), h, w, 1 if transposed else 0)
        return dst

    @staticmethod
    def forwardDCT2D(ctx, src):
        src = np.ascontiguousarray(src, np.float32)
        h, w = src.shape
        dst = np.empty((h, w), np.float32)
        ctx.call("jxl_stage_fdct2d", abi.fptr(src), abi.fptr(dst), h, w)
        return dst


class OpsinInverseMatrix:
    """J/color/OpsinInverseMatrix.java"""

    def __init__(self, matrix, opsin_bias, cbrt_opsin_bias):
        self.matrix = [float(v) for v in matrix]
        self.opsinBias = [float(v) for v in opsin_bias]
        self.cbrtOpsinBias = [float(v) for v in cbrt_opsin_bias]

    def invertXYB(self, ctx, buffer, intensityTarget):
        if len(buffer) < 3:
            raise ValueError("Can only XYB on 3 channels")  # OpsinInverseMatrix.java:106-107
        out = np.array(buffer, np.float32, order="C", copy=True)
        ctx.call("jxl_stage_xyb", _p3(out, C.c_float), out[0].size, abi.f9(*self.matrix), abi.f3(*self.opsinBias),
                 abi.f3(*self.cbrtOpsinBias), C.c_float(intensityTarget))
        return out


class LFCoefficients:
    """device part of J/frame/vardct/LFCoefficients.java (row f1): dequant, LF chroma-from-luma, adaptiveSmooth"""

    @staticmethod
    def dequantLFCoeff(ctx, lfQuant, scaledDequant, extraPrecision=0, xFactorLF=128, bFactorLF=128, adaptiveSmoothing=True,
                       baseCorrelationX=0.0, baseCorrelationB=1.0, colorFactor=84):
        """lfQuant: int32 [3][H][W] in X,Y,B order (lfQuant[cMap[i]]). Returns float32 [3][H][W]."""
        q = np.ascontiguousarray(lfQuant, np.int32)
        d = abi.make_lfquant_desc(q, scaledDequant, extraPrecision, xFactorLF, bFactorLF, adaptiveSmoothing)
        out = np.empty(q.shape, np.float32)
        ctx.call("jxl_stage_lf_dequant", C.byref(d), C.c_float(baseCorrelationX), C.c_float(baseCorrelationB), colorFactor,
                 _p3(out, C.c_float))
        return out


def performColorTransformsYCbCr(ctx, buffer):
    """YCbCr branch of JXLCodestreamDecoder.performColorTransforms (:270-281)"""
    out = np.array(buffer, np.float32, order="C", copy=True)
    ctx.call("jxl_stage_ycbcr", _p3(out, C.c_float), out[0].size)
    return out


def _one_colour(buffer):
    """A frame with ONE colour channel (Frame.getColorChannelCount, Frame.java:874-877: grey, not XYB, Modular). The reference's
    EPF distance still runs its channel loop three times but reads channel 0 every time (`i = colors == 1 ? 0 : c`,
    Frame.java:642,661): three copies of the plane through the three-channel kernels are that sum, term for term."""
    a = np.ascontiguousarray(buffer, np.float32)
    return a.ndim == 3 and a.shape[0] == 1


def performGabConvolution(ctx, buffer, gab1Weights, gab2Weights):
    """Frame.performGabConvolution (Frame.java:505-542)"""
    if _one_colour(buffer):  # channel 0 with ITS weights; the copies keep the kernels' three-plane interface
        a = np.ascontiguousarray(buffer, np.float32)
        return performGabConvolution(ctx, np.repeat(a, 3, axis=0), [gab1Weights[0]] * 3, [gab2Weights[0]] * 3)[:1]
    buf = _planes(buffer, np.float32)
    out = np.empty_like(buf)
    ctx.call("jxl_stage_gab", _p3(buf, C.c_float), _p3(out, C.c_float), buf.shape[1], buf.shape[2],
             abi.f3(*gab1Weights), abi.f3(*gab2Weights))
    return out


def epfInverseSigma(ctx, hfMultiplier, sharpness, globalScaleF, epfSharpLut):
    """inverse-sigma map of Frame.performEdgePreservingFilter (Frame.java:552-571); raises
    InvalidBitstreamException for sharpness outside 0..7 (:565-566)."""
    hf = np.ascontiguousarray(hfMultiplier, np.int32)
    sh = np.ascontiguousarray(sharpness, np.int32)
    out = np.empty(hf.shape, np.float32)
    ctx.call("jxl_stage_epf_sigma", abi.iptr(hf), abi.iptr(sh), hf.shape[0], hf.shape[1], C.c_float(globalScaleF),
             abi.f8(*epfSharpLut), abi.fptr(out))
    return out


def performEdgePreservingFilter(ctx, buffer, epfIterations, inverseSigma=None, invModularSigma=0.0,
                                epfChannelScale=(40.0, 5.0, 3.5), epfPass0SigmaScale=0.9, epfPass2SigmaScale=6.5,
                                epfBorderSadMul=2.0 / 3.0):
    """Frame.performEdgePreservingFilter iteration loop (Frame.java:583-635)"""
    if _one_colour(buffer):
        a = np.ascontiguousarray(buffer, np.float32)
        return performEdgePreservingFilter(ctx, np.repeat(a, 3, axis=0), epfIterations, inverseSigma, invModularSigma, epfChannelScale,
                                           epfPass0SigmaScale, epfPass2SigmaScale, epfBorderSadMul)[:1]
    buf = _planes(buffer, np.float32)
    out = np.empty_like(buf)
    sig = None
    if inverseSigma is not None:
        inverseSigma = np.ascontiguousarray(inverseSigma, np.float32)
        sig = abi.fptr(inverseSigma)
    ctx.call("jxl_stage_epf", _p3(buf, C.c_float), _p3(out, C.c_float), buf.shape[1], buf.shape[2], epfIterations, sig,
             C.c_float(invModularSigma), abi.f3(*epfChannelScale), C.c_float(epfPass0SigmaScale),
             C.c_float(epfPass2SigmaScale), C.c_float(epfBorderSadMul))
    return out


def transfer(ctx, x, tf, maxValue=0):
    """JXLImage.transferInPlace (+ ImageBuffer.castToIntWithMax when maxValue > 0)"""
    x = np.ascontiguousarray(x, np.float32)
    if maxValue > 0:
        out = np.empty(x.shape, np.int32)
        ctx.call("jxl_stage_transfer", abi.fptr(x), x.size, tf, maxValue, None, abi.iptr(out))
    else:
        out = np.empty(x.shape, np.float32)
        ctx.call("jxl_stage_transfer", abi.fptr(x), x.size, tf, 0, abi.fptr(out), None)
    return out


class Frame:
    """VarDCT side of J/frame/Frame.java: decodePassGroups tail (Frame.java:361-374), Gab, EPF and the
    colour transform of JXLCodestreamDecoder.performColorTransforms, as one device submission.

        fr = Frame(ctx, params, weights, woffs)
        fr.setLFGroup(g) ...            # HFMetadata / LFCoefficients side info, per LF group
        fr.putGroup(pass_, group, q)    # HFCoefficients.quantizedCoeffs of one (pass, group)
        planes = fr.decodeFrame()       # run + read back
    """

    def __init__(self, ctx, params, weights, woffs):
        self.ctx = ctx
        self.params = params
        self.width, self.height = params.width, params.height
        ctx.call("jxl_vardct_begin_frame", C.byref(params))
        weights = np.ascontiguousarray(weights, np.float32)
        woffs = np.ascontiguousarray(woffs, np.int32)
        ctx.call("jxl_vardct_set_weights", abi.fptr(weights), weights.size, abi.iptr(woffs))

    def setLFGroup(self, g):
        d = abi.make_lfgroup_desc(g)
        self.ctx.call("jxl_vardct_set_lfgroup", C.byref(d))

    def setLFGroupQuant(self, lfg_y, lfg_x, lfQuant, scaledDequant, extraPrecision=0, xFactorLF=128, bFactorLF=128,
                        adaptiveSmoothing=True):
        """row f1: hand over the integer LF image of an LF group instead of the dequantised floats"""
        q = [np.ascontiguousarray(lfQuant[c], np.int32) for c in range(3)]  # planes differ in size when chroma-subsampled
        d = abi.make_lfquant_desc(q, scaledDequant, extraPrecision, xFactorLF, bFactorLF, adaptiveSmoothing, lfg_y, lfg_x)
        self.ctx.call("jxl_vardct_set_lfgroup_lfquant", C.byref(d))

    @staticmethod
    def _rows(a, dt):
        """a 2-D plane as the ABI takes it: samples of a row consecutive, rows `stride` elements apart -- views with a row
        stride (a rectangle of a larger plane, page-locked or not) are passed as they are, anything else is made contiguous"""
        a = np.asarray(a)
        if a.dtype != dt or a.ndim != 2 or a.strides[1] != a.itemsize or a.strides[0] % a.itemsize or a.strides[0] < a.shape[1] * a.itemsize:
            a = np.ascontiguousarray(a, dt)
        return a

    def putGroup(self, pass_, group, q):
        q = [self._rows(a, np.int32) for a in q]
        pp = (C.POINTER(C.c_int32) * 3)(*[a.ctypes.data_as(C.POINTER(C.c_int32)) for a in q])
        strides = (C.c_int32 * 3)(*[a.strides[0] // 4 for a in q])
        self.ctx.call("jxl_vardct_put_group", pass_, group, pp, strides)
        self._keep = getattr(self, "_keep", []) + [q]  # aligned page-locked sources are read in place: keep them alive until run()

    def putGroupI16(self, pass_, group, q):
        """the int16 wire format (jxl_vardct_put_group_i16): the caller has checked that every |q| fits"""
        q = [self._rows(a, np.int16) for a in q]
        pp = (C.POINTER(C.c_int16) * 3)(*[a.ctypes.data_as(C.POINTER(C.c_int16)) for a in q])
        strides = (C.c_int32 * 3)(*[a.strides[0] // 2 for a in q])
        self.ctx.call("jxl_vardct_put_group_i16", pass_, group, pp, strides)
        self._keep = getattr(self, "_keep", []) + [q]  # aligned page-locked sources are read in place: keep them alive until run()

    def mapCoeffsI16(self, no_fill=False):
        """the frame's three coefficient planes as numpy views over the library's page-locked staging buffer
        (jxl_vardct_map_coeffs_i16): write the groups in place, then commitCoeffsI16(). no_fill: the planes are not
        zero-filled (jxl_vardct_map_coeffs_i16_ex, JXL_MAP_NO_FILL); commit then takes the list of written groups"""
        pp = (C.POINTER(C.c_int16) * 3)()
        strides = (C.c_int32 * 3)()
        if no_fill:
            self.ctx.call("jxl_vardct_map_coeffs_i16_ex", pp, strides, 1)
        else:
            self.ctx.call("jxl_vardct_map_coeffs_i16", pp, strides)
        p = self.params
        out = []
        for c in range(3):
            h, w = self.height >> p.jpeg_upsampling_y[c], self.width >> p.jpeg_upsampling_x[c]
            buf = (C.c_int16 * (h * w)).from_address(C.addressof(pp[c].contents))
            out.append(np.frombuffer(buf, dtype=np.int16).reshape(h, w))
        return out

    def commitCoeffsI16(self, written=None):
        """written: one flag per group (Frame group order) -- the groups whose rectangles were fully written; the others read
        as zero (jxl_vardct_commit_coeffs_i16_groups). None: everything in the mapped planes counts."""
        if written is None:
            self.ctx.call("jxl_vardct_commit_coeffs_i16")
        else:
            w = np.ascontiguousarray(written, np.uint8)
            self.ctx.call("jxl_vardct_commit_coeffs_i16_groups", w.ctypes.data_as(C.POINTER(C.c_uint8)), int(w.size))

    def run(self):
        """enqueue all stages (asynchronous)"""
        self.ctx.call("jxl_vardct_run")

    @staticmethod
    def runBatch(frames):
        """enqueue a batch of independent frames (one context each): jxl_vardct_run_batch -- the inverse-transform
        stage of all frames shares its launches; results equal those of frame.run() on every frame"""
        from ._lib import check
        hs = (C.c_void_p * len(frames))(*[f.ctx.h for f in frames])
        c0 = frames[0].ctx
        check(c0.h, c0.lib.jxl_vardct_run_batch(hs, len(frames)))

    def _out_array(self):
        es = self.ctx.lib.jxl_vardct_out_elem_size(self.ctx.h)
        p = self.params
        as_int = (p.stages & abi.STAGE_OUT) and p.out_format != abi.OUT_F32
        dt = {4: np.int32 if as_int else np.float32, 2: np.uint16, 1: np.uint8}[es]
        if as_int and p.out_format in (abi.OUT_RGB8, abi.OUT_RGB16):  # row f3: one pixel-interleaved buffer
            return np.empty((self.height, self.width, 3), dt)
        return np.empty((3, self.height, self.width), dt)

    def readOutput(self):
        out = self._out_array()
        if out.shape[-1] == 3 and out.ndim == 3 and out.shape[0] == self.height and out.dtype != np.float32 and \
                self.params.out_format in (abi.OUT_RGB8, abi.OUT_RGB16):
            pp = (C.c_void_p * 3)(out.ctypes.data, None, None)
        else:
            pp = (C.c_void_p * 3)(*[out[c].ctypes.data for c in range(3)])
        self.ctx.call("jxl_vardct_read_output", pp, self.width)
        return out

    def readOutputBegin(self, out=None):
        """queue the copy of the result to the host and return the destination array (jxl_vardct_read_output_begin); valid
        after readOutputWait(). `out`: a (page-locked) array of the result's shape to copy into"""
        if out is None:
            out = self._out_array()
        if self.params.out_format in (abi.OUT_RGB8, abi.OUT_RGB16) and (self.params.stages & abi.STAGE_OUT):
            pp = (C.c_void_p * 3)(out.ctypes.data, None, None)
        else:
            pp = (C.c_void_p * 3)(*[out[c].ctypes.data for c in range(3)])
        self.ctx.call("jxl_vardct_read_output_begin", pp, self.width)
        self._pending_out = out
        return out

    def readOutputWait(self):
        self.ctx.call("jxl_vardct_read_output_wait")
        out, self._pending_out = self._pending_out, None
        return out

    def decodeFrame(self):
        self.run()
        return self.readOutput()

    def lastLaunchCount(self):
        return self.ctx.lib.jxl_vardct_last_launch_count(self.ctx.h)

    def keepPlanes(self, height, width):
        """run, and keep the height x width window of the result on the device for the stages that follow decodeFrame
        (JXLCodestreamDecoder.java:628-637): -> ResidentPlanes"""
        self.run()
        self.ctx.call("jxl_planes_from_frame", height, width)
        return ResidentPlanes(self.ctx)

    @classmethod
    def from_synth(cls, ctx, frame, stages=None, via_groups=True):
        """feed a synth.make_vardct_frame dict through the boundary exactly as the Java host would:
        per LF group side info, per (pass, group) coefficient planes."""
        from . import synth
        p = abi.VarDCTParams.from_buffer_copy(frame["params"])
        if stages is not None:
            p.stages = stages
        fr = cls(ctx, p, frame["weights"], frame["woffs"])
        for g in frame["lfgroups"]:
            fr.setLFGroup(g)
        for grp in range(synth.num_groups(frame)):
            fr.putGroup(0, grp, synth.group_view(frame, grp))
        return fr


class PinnedArray:
    """numpy view over page-locked host memory from jxl_host_alloc (what a JNI caller wraps with NewDirectByteBuffer): copies
    to / from it are direct DMA, and put_group does not wait for them"""

    def __init__(self, lib, shape, dtype):
        self.lib = lib
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        self.ptr = lib.jxl_host_alloc(max(n, 1))
        if not self.ptr:
            raise MemoryError("jxl_host_alloc(%d)" % n)
        buf = (C.c_char * max(n, 1)).from_address(self.ptr)
        self.array = np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)

    def free(self):
        if self.ptr:
            self.array = None
            self.lib.jxl_host_free(self.ptr)
            self.ptr = None


class ResidentPlanes:
    """the frame's three colour planes on the device between decodeFrame and the blend: Frame.upsample (Frame.java:217-260),
    initializeNoise + synthesizeNoise (:748-831), performColorTransforms (JXLCodestreamDecoder.java:256-276), in the
    reference's order, with the host hook (download / upload) only where the reference's host-side stages (patches,
    splines, saveBeforeCT) need the samples."""

    def __init__(self, ctx):
        self.ctx = ctx

    @classmethod
    def upload(cls, ctx, planes):
        pl = _planes(planes, np.float32)
        ctx.call("jxl_planes_upload", _p3(pl, C.c_float), pl[0].shape[0], pl[0].shape[1])
        return cls(ctx)

    @property
    def shape(self):
        h, w = C.c_int32(), C.c_int32()
        self.ctx.call("jxl_planes_shape", C.byref(h), C.byref(w))
        return h.value, w.value

    def upsample(self, k, upWeights):
        w = np.ascontiguousarray(upWeights, np.float32)
        assert w.size == k * k * 25
        self.ctx.call("jxl_planes_upsample", k, abi.fptr(w))

    def noise(self, groupDim, seed0, lut, baseCorrelationX, baseCorrelationB):
        lut = np.ascontiguousarray(lut, np.float32)
        assert lut.size == 8
        self.ctx.call("jxl_planes_noise", groupDim, C.c_uint64(seed0), abi.fptr(lut), C.c_float(baseCorrelationX),
                      C.c_float(baseCorrelationB))

    def invertXYB(self, matrix, opsin_bias, cbrt_opsin_bias, intensityTarget):
        m = OpsinInverseMatrix(matrix, opsin_bias, cbrt_opsin_bias)
        self.ctx.call("jxl_planes_xyb", abi.f9(*m.matrix), abi.f3(*m.opsinBias), abi.f3(*m.cbrtOpsinBias), C.c_float(intensityTarget))

    def ycbcr(self):
        self.ctx.call("jxl_planes_ycbcr")

    def download(self):
        h, w = self.shape
        out = np.empty((3, h, w), np.float32)
        self.ctx.call("jxl_planes_download", _p3(out, C.c_float))
        return out

    def replace(self, planes):
        pl = _planes(planes, np.float32)
        self.ctx.call("jxl_planes_upload", _p3(pl, C.c_float), pl[0].shape[0], pl[0].shape[1])


class ModularChannel:
    """J/frame/modular/ModularChannel.java (squeeze part)"""

    @staticmethod
    def inverseHorizontalSqueeze(ctx, orig, res):
        orig = np.ascontiguousarray(orig, np.int32)
        res = np.ascontiguousarray(res, np.int32)
        h, aw = orig.shape
        if res.shape[0] != h:
            raise ValueError("Corrupted squeeze transform")  # ModularChannel.java:363-366
        out = np.empty((h, aw + res.shape[1]), np.int32)
        ctx.call("jxl_stage_inv_hsqueeze", abi.iptr(orig), aw, abi.iptr(res), res.shape[1], h, abi.iptr(out))
        return out

    @staticmethod
    def inverseVerticalSqueeze(ctx, orig, res):
        orig = np.ascontiguousarray(orig, np.int32)
        res = np.ascontiguousarray(res, np.int32)
        ah, w = orig.shape
        if res.shape[1] != w:
            raise RuntimeError("Corrupted squeeze transform")  # ModularChannel.java:391-394 (IllegalStateException)
        out = np.empty((ah + res.shape[0], w), np.int32)
        ctx.call("jxl_stage_inv_vsqueeze", abi.iptr(orig), ah, abi.iptr(res), res.shape[0], w, abi.iptr(out))
        return out


class ModularStream:
    """Transform part of J/frame/modular/ModularStream.java: the channel list as decoded by the host
    (averages + residuals) plus the squeeze parameters; applyTransforms() undoes Squeeze (and RCT)."""

    def __init__(self, ctx, channels, squeezeParams, rctType=-1, rctBegin=0):
        self.ctx = ctx
        self.channels = [np.ascontiguousarray(c, np.int32) for c in channels]
        self.sp = [tuple(int(v) for v in s) for s in squeezeParams]
        self.rctType, self.rctBegin = rctType, rctBegin
        self.transformed = False
        self._begun = False

    @staticmethod
    def defaultSqueezeParams(shapes, nbMeta=0):
        """ModularStream.java:110-131 through the C-ABI (shapes: list of (h, w))"""
        from ._lib import load
        ws = np.array([s[1] for s in shapes], np.int32)
        hs = np.array([s[0] for s in shapes], np.int32)
        out = (abi.SqueezeParam * 64)()
        n = load().jxl_modular_default_squeeze_params(abi.iptr(ws), abi.iptr(hs), len(shapes), nbMeta, out, 64)
        if n < 0:
            raise ValueError("status %d" % n)
        return [out[i].as_tuple() for i in range(n)]

    @staticmethod
    def squeezedShapes(shapes, sp):
        from ._lib import load
        ws = np.array([s[1] for s in shapes], np.int32)
        hs = np.array([s[0] for s in shapes], np.int32)
        cap = len(shapes) + sum(p[3] for p in sp) + 1
        ow, oh = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        n = load().jxl_modular_squeezed_shapes(abi.iptr(ws), abi.iptr(hs), len(shapes), abi.make_squeeze_params(sp), len(sp),
                                               abi.iptr(ow), abi.iptr(oh), cap)
        if n < 0:
            raise ValueError("status %d" % n)
        return [(int(oh[i]), int(ow[i])) for i in range(n)]

    def begin(self):
        ca = abi.make_channels(self.channels)
        self.ctx.call("jxl_modular_begin", ca, len(self.channels), abi.make_squeeze_params(self.sp), len(self.sp),
                      self.rctType, self.rctBegin)
        self._begun = True

    def run(self):
        if not self._begun:
            self.begin()
        self.ctx.call("jxl_modular_run")

    def getDecodedBuffer(self):
        lib, h = self.ctx.lib, self.ctx.h
        outs = []
        for i in range(lib.jxl_modular_out_count(h)):
            w, hh = C.c_int32(), C.c_int32()
            check(h, lib.jxl_modular_out_shape(h, i, C.byref(w), C.byref(hh)))
            a = np.empty((hh.value, w.value), np.int32)
            self.ctx.call("jxl_modular_read_channel", i, abi.iptr(a) if a.size else None)
            outs.append(a)
        return outs

    def applyTransforms(self):
        """ModularStream.applyTransforms (ModularStream.java:224-380): idempotent like the reference"""
        if self.transformed:
            return self.channels
        self.transformed = True
        self.run()
        self.channels = self.getDecodedBuffer()
        return self.channels


def rct(ctx, v, rctType):
    """RCT branch of ModularStream.applyTransforms (ModularStream.java:255-326)"""
    out = np.array(v, np.int32, order="C", copy=True)
    pp = (C.POINTER(C.c_int32) * 3)(*[abi.iptr(out[c]) for c in range(3)])
    ctx.call("jxl_stage_rct", pp, out[0].size, rctType)
    return out


def modularToFloat(ctx, a, b, scale):
    """Frame.decodeFrame modular -> float buffer (Frame.java:437-448)"""
    a = np.ascontiguousarray(a, np.int32)
    out = np.empty(a.shape, np.float32)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, np.int32)
        bp = abi.iptr(b)
    ctx.call("jxl_stage_modular_to_float", abi.iptr(a), bp, a.size, C.c_float(scale), abi.fptr(out))
    return out


# ---- rows f4 / f3: the pixel-domain functions either side of the colour transform -------------------------
def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def invertSubsampling(ctx, channel, xShift, yShift):
    """Frame.invertSubsampling (Frame.java:681-723) for one channel"""
    a = np.ascontiguousarray(channel, np.float32)
    h, w = a.shape
    out = np.empty((h << yShift, w << xShift), np.float32)
    ctx.call("jxl_stage_chroma_upsample", abi.fptr(a), h, w, xShift, yShift, abi.fptr(out))
    return out


def getUpWeights(k, packed):
    """ImageHeader.getUpWeights (ImageHeader.java:441-470) for one k: [k][k][5][5]"""
    from ._lib import load
    packed = np.ascontiguousarray(packed, np.float32)
    need = {2: 15, 4: 55, 8: 210}.get(k)
    if need is None or packed.size != need:
        raise ValueError("k must be 2, 4 or 8 with 15 / 55 / 210 coefficients")
    out = np.empty((k, k, 5, 5), np.float32)
    st = load().jxl_upsampling_weights(k, abi.fptr(packed), abi.fptr(out))
    if st:
        raise ValueError("status %d" % st)
    return out


def performUpsampling(ctx, channel, k, upWeights):
    """Frame.performUpsampling (Frame.java:217-260)"""
    if k == 1:
        return channel
    a = np.ascontiguousarray(channel, np.float32)
    wts = np.ascontiguousarray(upWeights, np.float32)
    h, w = a.shape
    out = np.empty((h * k, w * k), np.float32)
    ctx.call("jxl_stage_upsample", abi.fptr(a), h, w, k, abi.fptr(wts), abi.fptr(out))
    return out


def initializeNoise(ctx, height, width, seed0, groupDim=256, colors=3):
    """Frame.initializeNoise (Frame.java:748-788)"""
    out = np.empty((colors, height, width), np.float32)
    pp = (C.POINTER(C.c_float) * 3)(*[abi.fptr(out[c]) for c in range(colors)])
    ctx.call("jxl_stage_noise_init", height, width, groupDim, C.c_uint64(seed0 & 0xFFFFFFFFFFFFFFFF), colors, pp)
    return out


def synthesizeNoise(ctx, planes, noise, lut, baseCorrelationX, baseCorrelationB):
    """Frame.synthesizeNoise (Frame.java:790-831); returns new planes"""
    out = np.array(planes, np.float32, order="C", copy=True)
    nz = np.ascontiguousarray(noise, np.float32)
    lut = np.ascontiguousarray(lut, np.float32)
    pp = (C.POINTER(C.c_float) * 3)(*[abi.fptr(out[c]) for c in range(3)])
    pn = (C.POINTER(C.c_float) * 3)(*[abi.fptr(nz[c]) for c in range(3)])
    ctx.call("jxl_stage_noise_add", pp, pn, out[0].size, abi.fptr(lut), C.c_float(baseCorrelationX), C.c_float(baseCorrelationB))
    return out


def blend(ctx, mode, canvas, frame, ref, rect, frameAlpha=None, refAlpha=None, isAlpha=False, hasExtra=False, clamp=False,
          premult=False):
    """inner switch of JXLCodestreamDecoder.blendBuffers (JXLCodestreamDecoder.java:285-422); rect =
    (h, w, canvas_y, canvas_x, frame_y, frame_x, ref_y, ref_x); returns the updated canvas"""
    is_int = canvas.dtype == np.int32
    dt = np.int32 if is_int else np.float32
    cv = np.array(canvas, dt, order="C", copy=True)
    fr = np.ascontiguousarray(frame, dt) if frame is not None else None
    rf = np.ascontiguousarray(ref, dt) if ref is not None else None
    fa = np.ascontiguousarray(frameAlpha, np.float32) if frameAlpha is not None else None
    ra = np.ascontiguousarray(refAlpha, np.float32) if refAlpha is not None else None
    flags = (1 if isAlpha else 0) | (2 if hasExtra else 0) | (4 if clamp else 0) | (8 if premult else 0)
    r = abi.BlendRect(*[int(v) for v in rect])
    fh, fw = fr.shape if fr is not None else (fa.shape if fa is not None else (0, 0))
    rh, rw = rf.shape if rf is not None else (ra.shape if ra is not None else (0, 0))
    ctx.call("jxl_stage_blend", mode, flags, 1 if is_int else 0, _vp(cv), cv.shape[0], cv.shape[1], _vp(fr), fh, fw,
             _vp(rf), rh, rw, abi.fptr(fa) if fa is not None else None, abi.fptr(ra) if ra is not None else None, C.byref(r))
    return cv


def transposeBuffer(ctx, src, orientation):
    """JXLCodestreamDecoder.transposeBuffer (JXLCodestreamDecoder.java:43-184)"""
    if src.dtype not in (np.int32, np.float32):
        raise TypeError("int32 or float32 planes")
    a = np.ascontiguousarray(src)
    h, w = a.shape
    out = np.empty((w, h) if orientation > 4 else (h, w), a.dtype)
    ctx.call("jxl_stage_orient", _vp(a), h, w, orientation, _vp(out))
    return out


def packSamples(ctx, planes, bitDepth, alpha=None, premultiplied=False, taggedDepth=None, bigEndian=False):
    """PNGWriter ctor tail + writeIDAT sample order (PNGWriter.java:79-111, 191-203): planes = 1 or 3 colour planes
    (int32 or float32 each), optional alpha plane; returns [h][w][channels] uint8 / uint16"""
    pl = [np.ascontiguousarray(p) for p in planes] + ([np.ascontiguousarray(alpha)] if alpha is not None else [])
    h, w = pl[0].shape
    p = abi.PackParams()
    p.height, p.width, p.n_color, p.has_alpha = h, w, len(planes), 1 if alpha is not None else 0
    p.premultiplied, p.bit_depth, p.big_endian = int(bool(premultiplied)), bitDepth, int(bool(bigEndian))
    for i, a in enumerate(pl):
        if a.dtype not in (np.int32, np.float32) or a.shape != (h, w):
            raise TypeError("planes must be int32 or float32 of one shape")
        p.is_int[i] = 1 if a.dtype == np.int32 else 0
        p.tagged_depth[i] = (taggedDepth[i] if taggedDepth is not None else bitDepth)
    out = np.empty((h, w, len(pl)), np.uint8 if bitDepth == 8 else np.uint16)
    pp = (C.c_void_p * 4)(*([_vp(a) for a in pl] + [None] * (4 - len(pl))))
    ctx.call("jxl_stage_pack", pp, C.byref(p), _vp(out))
    return out

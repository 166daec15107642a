"""ctypes wrapper of the CPU oracle (oracle/libjxl_oracle.so).

TEST INFRASTRUCTURE ONLY -- imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py; never by the product package jxlatte_amd. PARITY UNPINNED
(see oracle/jxl_oracle.h).
"""
import ctypes as C
import os
import subprocess

import numpy as np

from jxlatte_amd import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libjxl_oracle.so")


def build(force=False):
    """compile the oracle with gcc (oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in ("jxl_oracle.c", "jxl_oracle_post.c", "jxl_oracle.h")]
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


class OrcFrame(C.Structure):
    _fields_ = [
        ("p", abi.VarDCTParams), ("weights", C.POINTER(C.c_float)), ("woffs", C.POINTER(C.c_int32)),
        ("n_lfg", C.c_int32), ("lfg", C.POINTER(abi.LFGroupDesc)),
        ("coeff", C.POINTER(C.c_int32) * 3), ("threads", C.c_int32),
    ]


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.orc_cosine_lut.restype = C.POINTER(C.c_float)
        L.orc_cosine_lut.argtypes = [C.c_int]
        L.orc_vardct_frame_run.restype = C.c_int32
        L.orc_epf_sigma.restype = C.c_int32
        L.orc_rct.restype = C.c_int32
        L.orc_modular_apply.restype = C.c_int32
        L.orc_default_squeeze_params.restype = C.c_int32
        L.orc_squeezed_shapes.restype = C.c_int32
        for f in ("orc_upsampling_weights", "orc_blend", "orc_orient", "orc_pack"):
            getattr(L, f).restype = C.c_int32
        _lib = L
    return _lib


def cosine_lut(l):
    s = 1 << l
    if s == 1:
        return np.zeros((0, 1), np.float32)
    p = lib().orc_cosine_lut(l)
    return np.ctypeslib.as_array(p, shape=((s - 1) * s,)).reshape(s - 1, s).copy()


def idct1d(src):
    src = np.ascontiguousarray(src, np.float32)
    n = src.shape[0]
    dst = np.empty(n, np.float32)
    lib().orc_idct1d(abi.fptr(src), abi.fptr(dst), C.c_int(int(np.log2(n))), C.c_int(n))
    return dst


def fdct1d(src):
    src = np.ascontiguousarray(src, np.float32)
    n = src.shape[0]
    dst = np.empty(n, np.float32)
    lib().orc_fdct1d(abi.fptr(src), abi.fptr(dst), C.c_int(int(np.log2(n))), C.c_int(n))
    return dst


def idct2d(src, transposed=False):
    src = np.ascontiguousarray(src, np.float32)
    h, w = src.shape
    dst = np.empty((w, h) if transposed else (h, w), np.float32)
    lib().orc_idct2d(abi.fptr(src), C.c_int64(w), abi.fptr(dst), C.c_int64(dst.shape[1]), C.c_int(h), C.c_int(w),
                     C.c_int(1 if transposed else 0))
    return dst


def fdct2d(src):
    src = np.ascontiguousarray(src, np.float32)
    h, w = src.shape
    dst = np.empty((h, w), np.float32)
    lib().orc_fdct2d(abi.fptr(src), C.c_int64(w), abi.fptr(dst), C.c_int64(w), C.c_int(h), C.c_int(w))
    return dst


def _planes3(a, ctype):
    arr = (C.POINTER(ctype) * 3)()
    for c in range(3):
        arr[c] = abi.ptr(a[c], ctype)
    return arr


def vardct_frame(frame, stages=None, threads=0):
    """run the oracle on a synth frame dict; returns float32 [3][H][W] (or int32 for int output)."""
    p = abi.VarDCTParams.from_buffer_copy(frame["params"])
    if stages is not None:
        p.stages = stages
    H, W = p.height, p.width
    f = OrcFrame()
    f.p = p
    f.weights = abi.fptr(frame["weights"])
    f.woffs = abi.iptr(frame["woffs"])
    descs = (abi.LFGroupDesc * len(frame["lfgroups"]))()
    for i, g in enumerate(frame["lfgroups"]):
        descs[i] = abi.make_lfgroup_desc(g)
    f.n_lfg = len(frame["lfgroups"])
    f.lfg = descs
    coeff = frame["coeff"]
    assert coeff.dtype == np.int32 and coeff.shape == (3, H, W) and coeff.flags["C_CONTIGUOUS"]
    for c in range(3):
        f.coeff[c] = abi.iptr(coeff[c])
    f.threads = threads
    as_int = (p.stages & abi.STAGE_OUT) and p.out_format != abi.OUT_F32
    out = np.zeros((3, H, W), np.int32 if as_int else np.float32)
    outp = (C.c_void_p * 3)()
    for c in range(3):
        outp[c] = out[c].ctypes.data
    st = lib().orc_vardct_frame_run(C.byref(f), outp)
    if st != 0:
        raise RuntimeError("oracle vardct status %d" % st)
    return out


def gab(planes, w1, w2):
    planes = np.ascontiguousarray(planes, np.float32)
    _, h, w = planes.shape
    out = np.empty_like(planes)
    lib().orc_gab(_planes3(planes, C.c_float), _planes3(out, C.c_float), C.c_int(h), C.c_int(w),
                  abi.f3(*w1), abi.f3(*w2))
    return out


def gab1(plane, w1, w2):
    """performGabConvolution of a one-colour frame (colors == 1: the channel loop runs once, Frame.java:518)"""
    p = np.ascontiguousarray(plane, np.float32)
    return gab(np.stack([p, p, p]), [w1] * 3, [w2] * 3)[0]


def epf1(plane, iterations, inv_sigma, inv_sigma_modular, channel_scale, pass0, pass2, border_sad_mul):
    """performEdgePreservingFilter of a one-colour frame: epfDistance1/2 still add three rounds, each reading buffer[0]
    (`int i = colors == 1 ? 0 : c`, Frame.java:642,661) with that round's channel scale -- which is what the three-plane
    restatement computes when all three planes ARE channel 0; sumChannels / outputBuffers have one entry (Frame.java:600,625)."""
    p = np.ascontiguousarray(plane, np.float32)
    return epf(np.stack([p, p, p]), iterations, inv_sigma, inv_sigma_modular, channel_scale, pass0, pass2, border_sad_mul)[0]


def epf_sigma(hf_mul, sharpness, global_scale_f, sharp_lut):
    hf_mul = np.ascontiguousarray(hf_mul, np.int32)
    sharpness = np.ascontiguousarray(sharpness, np.int32)
    bh, bw = hf_mul.shape
    out = np.empty((bh, bw), np.float32)
    st = lib().orc_epf_sigma(abi.iptr(hf_mul), abi.iptr(sharpness), C.c_int(bh), C.c_int(bw),
                             C.c_float(global_scale_f), abi.f8(*sharp_lut), abi.fptr(out))
    if st != 0:
        raise ValueError("oracle epf_sigma status %d" % st)
    return out


def epf(planes, iterations, inv_sigma, inv_sigma_modular, channel_scale, pass0, pass2, border_sad_mul):
    planes = np.ascontiguousarray(planes, np.float32)
    _, h, w = planes.shape
    out = np.empty_like(planes)
    sig = None
    if inv_sigma is not None:
        inv_sigma = np.ascontiguousarray(inv_sigma, np.float32)
        sig = abi.fptr(inv_sigma)
    lib().orc_epf(_planes3(planes, C.c_float), _planes3(out, C.c_float), C.c_int(h), C.c_int(w),
                  C.c_int(iterations), sig, C.c_float(inv_sigma_modular), abi.f3(*channel_scale),
                  C.c_float(pass0), C.c_float(pass2), C.c_float(border_sad_mul))
    return out


def xyb(planes, matrix, opsin_bias, cbrt_opsin_bias, intensity_target):
    out = np.array(planes, np.float32, order="C", copy=True)
    n = out[0].size
    lib().orc_xyb(_planes3(out, C.c_float), C.c_int64(n), abi.f9(*matrix), abi.f3(*opsin_bias),
                  abi.f3(*cbrt_opsin_bias), C.c_float(intensity_target))
    return out


def lf_dequant(lf_quant, scaled_dequant, extra_precision=0, x_factor_lf=128, b_factor_lf=128, adaptive_smoothing=True,
               base_corr_x=0.0, base_corr_b=1.0, color_factor=84):
    q = np.ascontiguousarray(lf_quant, np.int32)
    d = abi.make_lfquant_desc(q, scaled_dequant, extra_precision, x_factor_lf, b_factor_lf, adaptive_smoothing)
    out = np.empty(q.shape, np.float32)
    lib().orc_lf_dequant(C.byref(d), C.c_float(base_corr_x), C.c_float(base_corr_b), C.c_int32(color_factor), _planes3(out, C.c_float))
    return out


def ycbcr(planes):
    out = np.array(planes, np.float32, order="C", copy=True)
    lib().orc_ycbcr(_planes3(out, C.c_float), C.c_int64(out[0].size))
    return out


def transfer(x, tf, max_value=0):
    x = np.ascontiguousarray(x, np.float32)
    if max_value > 0:
        out = np.empty(x.shape, np.int32)
        lib().orc_transfer(abi.fptr(x), C.c_int64(x.size), C.c_int(tf), C.c_int(max_value), None, abi.iptr(out))
    else:
        out = np.empty(x.shape, np.float32)
        lib().orc_transfer(abi.fptr(x), C.c_int64(x.size), C.c_int(tf), C.c_int(0), abi.fptr(out), None)
    return out


def inv_hsqueeze(avg, res):
    avg = np.ascontiguousarray(avg, np.int32)
    res = np.ascontiguousarray(res, np.int32)
    h, aw = avg.shape
    rw = res.shape[1]
    out = np.empty((h, aw + rw), np.int32)
    lib().orc_inv_hsqueeze(abi.iptr(avg), C.c_int(aw), abi.iptr(res), C.c_int(rw), C.c_int(h), abi.iptr(out))
    return out


def inv_vsqueeze(avg, res):
    avg = np.ascontiguousarray(avg, np.int32)
    res = np.ascontiguousarray(res, np.int32)
    ah, w = avg.shape
    rh = res.shape[0]
    out = np.empty((ah + rh, w), np.int32)
    lib().orc_inv_vsqueeze(abi.iptr(avg), C.c_int(ah), abi.iptr(res), C.c_int(rh), C.c_int(w), abi.iptr(out))
    return out


def fwd_hsqueeze(img):
    img = np.ascontiguousarray(img, np.int32)
    h, w = img.shape
    avg = np.empty((h, (w + 1) // 2), np.int32)
    res = np.empty((h, w // 2), np.int32)
    lib().orc_fwd_hsqueeze(abi.iptr(img), C.c_int(h), C.c_int(w), abi.iptr(avg), abi.iptr(res))
    return avg, res


def fwd_vsqueeze(img):
    img = np.ascontiguousarray(img, np.int32)
    h, w = img.shape
    avg = np.empty(((h + 1) // 2, w), np.int32)
    res = np.empty((h // 2, w), np.int32)
    lib().orc_fwd_vsqueeze(abi.iptr(img), C.c_int(h), C.c_int(w), abi.iptr(avg), abi.iptr(res))
    return avg, res


def rct(v, rct_type):
    out = np.array(v, np.int32, order="C", copy=True)
    n = out[0].size
    pp = (C.POINTER(C.c_int32) * 3)(*[abi.iptr(out[c]) for c in range(3)])
    st = lib().orc_rct(pp, C.c_int64(n), C.c_int(rct_type))
    if st != 0:
        raise ValueError("oracle rct status %d" % st)
    return out


def modular_to_float(a, b, scale):
    a = np.ascontiguousarray(a, np.int32)
    out = np.empty(a.shape, np.float32)
    bp = None
    if b is not None:
        b = np.ascontiguousarray(b, np.int32)
        bp = abi.iptr(b)
    lib().orc_modular_to_float(abi.iptr(a), bp, C.c_int64(a.size), C.c_float(scale), abi.fptr(out))
    return out


def default_squeeze_params(shapes, nb_meta=0):
    """shapes: list of (h, w). returns list of (horizontal, in_place, begin_c, num_c)."""
    ws = np.array([s[1] for s in shapes], np.int32)
    hs = np.array([s[0] for s in shapes], np.int32)
    out = (abi.SqueezeParam * 64)()
    n = lib().orc_default_squeeze_params(abi.iptr(ws), abi.iptr(hs), C.c_int32(len(shapes)), C.c_int32(nb_meta), out, C.c_int32(64))
    if n < 0:
        raise ValueError(n)
    return [out[i].as_tuple() for i in range(n)]


def squeezed_shapes(shapes, sp):
    ws = np.array([s[1] for s in shapes], np.int32)
    hs = np.array([s[0] for s in shapes], np.int32)
    cap = len(shapes) + sum(p[3] for p in sp) + 1
    ow = np.zeros(cap, np.int32)
    oh = np.zeros(cap, np.int32)
    spa = abi.make_squeeze_params(sp)
    n = lib().orc_squeezed_shapes(abi.iptr(ws), abi.iptr(hs), C.c_int32(len(shapes)), spa, C.c_int32(len(sp)),
                                  abi.iptr(ow), abi.iptr(oh), C.c_int32(cap))
    if n < 0:
        raise ValueError(n)
    return [(int(oh[i]), int(ow[i])) for i in range(n)]


def modular_apply(chans, sp, rct_type=-1, rct_begin=0, out_shapes=None):
    """chans: encoded channel list (2-D int32 arrays); returns the channels after the inverse."""
    chans = [np.ascontiguousarray(c, np.int32) for c in chans]
    if out_shapes is None:
        out_shapes = inverse_shapes([c.shape for c in chans], sp)
    outs = [np.zeros(s, np.int32) for s in out_shapes]
    ca = abi.make_channels(chans)
    oa = abi.make_channels(outs)
    spa = abi.make_squeeze_params(sp)
    st = lib().orc_modular_apply(ca, C.c_int32(len(chans)), spa, C.c_int32(len(sp)), C.c_int32(rct_type),
                                 C.c_int32(rct_begin), oa, C.c_int32(len(outs)))
    if st != 0:
        raise ValueError("oracle modular_apply status %d" % st)
    return outs


def inverse_shapes(shapes, sp):
    """channel shapes after undoing the squeeze steps (pure bookkeeping, ModularStream.java:229-254)."""
    shapes = [tuple(s) for s in shapes]
    for (horiz, in_place, begin, num) in reversed(sp):
        end = begin + num - 1
        offset = end + 1 if in_place else len(shapes) + begin - end - 1
        for c in range(begin, end + 1):
            r = offset + c - begin
            if horiz:
                shapes[c] = (shapes[c][0], shapes[c][1] + shapes[r][1])
            else:
                shapes[c] = (shapes[c][0] + shapes[r][0], shapes[c][1])
        del shapes[offset:offset + num]
    return shapes


# ---- rows f4 / f3 (jxl_oracle_post.c) ----------------------------------------------------------------------
def _vp(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


def chroma_upsample(channel, x_shift, y_shift):
    a = np.ascontiguousarray(channel, np.float32)
    h, w = a.shape
    out = np.empty((h << y_shift, w << x_shift), np.float32)
    lib().orc_chroma_upsample(abi.fptr(a), C.c_int(h), C.c_int(w), C.c_int(x_shift), C.c_int(y_shift), abi.fptr(out))
    return out


def upsampling_weights(k, packed):
    packed = np.ascontiguousarray(packed, np.float32)
    out = np.empty((k, k, 5, 5), np.float32)
    st = lib().orc_upsampling_weights(C.c_int(k), abi.fptr(packed), abi.fptr(out))
    assert st == 0, st
    return out


def upsample(channel, k, weights):
    a = np.ascontiguousarray(channel, np.float32)
    wts = np.ascontiguousarray(weights, np.float32)
    h, w = a.shape
    out = np.empty((h * k, w * k), np.float32)
    lib().orc_upsample(abi.fptr(a), C.c_int(h), C.c_int(w), C.c_int(k), abi.fptr(wts), abi.fptr(out))
    return out


def noise_init(h, w, seed0, group_dim=256, colors=3):
    out = np.empty((colors, h, w), np.float32)
    pp = (C.POINTER(C.c_float) * 3)(*[abi.fptr(out[c]) for c in range(colors)])
    lib().orc_noise_init(C.c_int(h), C.c_int(w), C.c_int(group_dim), C.c_uint64(seed0 & 0xFFFFFFFFFFFFFFFF), C.c_int(colors), pp)
    return out


def noise_add(planes, noise, lut, base_corr_x, base_corr_b):
    out = np.array(planes, np.float32, order="C", copy=True)
    nz = np.ascontiguousarray(noise, np.float32)
    lut = np.ascontiguousarray(lut, np.float32)
    pp = (C.POINTER(C.c_float) * 3)(*[abi.fptr(out[c]) for c in range(3)])
    pn = (C.POINTER(C.c_float) * 3)(*[abi.fptr(nz[c]) for c in range(3)])
    lib().orc_noise_add(pp, pn, C.c_int64(out[0].size), abi.fptr(lut), C.c_float(base_corr_x), C.c_float(base_corr_b))
    return out


def blend(mode, canvas, frame, ref, rect, frame_alpha=None, ref_alpha=None, is_alpha=False, has_extra=False, clamp=False,
          premult=False):
    is_int = canvas.dtype == np.int32
    dt = np.int32 if is_int else np.float32
    cv = np.array(canvas, dt, order="C", copy=True)
    fr = np.ascontiguousarray(frame, dt) if frame is not None else None
    rf = np.ascontiguousarray(ref, dt) if ref is not None else None
    fa = np.ascontiguousarray(frame_alpha, np.float32) if frame_alpha is not None else None
    ra = np.ascontiguousarray(ref_alpha, np.float32) if ref_alpha is not None else None
    flags = (1 if is_alpha else 0) | (2 if has_extra else 0) | (4 if clamp else 0) | (8 if premult else 0)
    r = abi.BlendRect(*[int(v) for v in rect])
    fh, fw = fr.shape if fr is not None else (fa.shape if fa is not None else (0, 0))
    rh, rw = rf.shape if rf is not None else (ra.shape if ra is not None else (0, 0))
    st = lib().orc_blend(C.c_int(mode), C.c_uint32(flags), C.c_int(1 if is_int else 0), _vp(cv), C.c_int(cv.shape[0]),
                         C.c_int(cv.shape[1]), _vp(fr), C.c_int(fh), C.c_int(fw), _vp(rf), C.c_int(rh), C.c_int(rw),
                         abi.fptr(fa) if fa is not None else None, abi.fptr(ra) if ra is not None else None, C.byref(r))
    return st, cv


def orient(src, orientation):
    a = np.ascontiguousarray(src)
    h, w = a.shape
    out = np.empty((w, h) if orientation > 4 else (h, w), a.dtype)
    st = lib().orc_orient(_vp(a), C.c_int(h), C.c_int(w), C.c_int(orientation), _vp(out))
    assert st == 0, st
    return out


def pack(planes, bit_depth, alpha=None, premultiplied=False, tagged_depth=None, big_endian=False):
    pl = [np.ascontiguousarray(p) for p in planes] + ([np.ascontiguousarray(alpha)] if alpha is not None else [])
    h, w = pl[0].shape
    p = abi.PackParams()
    p.height, p.width, p.n_color, p.has_alpha = h, w, len(planes), 1 if alpha is not None else 0
    p.premultiplied, p.bit_depth, p.big_endian = int(bool(premultiplied)), bit_depth, int(bool(big_endian))
    for i, a in enumerate(pl):
        p.is_int[i] = 1 if a.dtype == np.int32 else 0
        p.tagged_depth[i] = (tagged_depth[i] if tagged_depth is not None else bit_depth)
    out = np.empty((h, w, len(pl)), np.uint8 if bit_depth == 8 else np.uint16)
    pp = (C.c_void_p * 4)(*([_vp(a) for a in pl] + [None] * (4 - len(pl))))
    st = lib().orc_pack(pp, C.byref(p), _vp(out))
    assert st == 0, st
    return out

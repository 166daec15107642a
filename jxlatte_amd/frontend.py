"""ctypes binding of libjxlatte_frontend.so (include/jxlatte_frontend.h): the host-side JPEG XL bitstream front-end.

Row f2 of the scope table: what the Java host does before the transform stage (container, headers, entropy decoding,
MA trees, TOC sections). CPU code, no GPU needed; the frame-level inverse Squeeze / RCT are delegated through hooks."""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.path.join(HERE, "libjxlatte_frontend.so")
MAX_EXTRA = 16
i32, f32 = C.c_int32, C.c_float


def _arr(t, n):
    return t * n


class ImageInfo(C.Structure):
    _fields_ = [("width", i32), ("height", i32), ("level", i32), ("orientation", i32),
                ("bits_per_sample", i32), ("exp_bits", i32), ("modular_16bit", i32), ("num_extra", i32), ("xyb_encoded", i32),
                ("colour_space", i32), ("white_point", i32), ("primaries", i32), ("transfer", i32), ("rendering_intent", i32),
                ("use_icc", i32), ("white_xy", _arr(f32, 2)), ("prim_xy", _arr(f32, 6)),
                ("intensity_target", f32), ("min_nits", f32), ("linear_below", f32), ("relative_to_max_display", i32),
                ("opsin_matrix", _arr(f32, 9)), ("opsin_bias", _arr(f32, 3)), ("quant_bias", _arr(f32, 3)),
                ("quant_bias_numerator", f32), ("have_animation", i32), ("have_preview", i32), ("custom_up", _arr(i32, 3)),
                ("ec_type", _arr(i32, MAX_EXTRA)), ("ec_bits", _arr(i32, MAX_EXTRA)), ("ec_exp_bits", _arr(i32, MAX_EXTRA)),
                ("ec_dim_shift", _arr(i32, MAX_EXTRA)), ("ec_alpha_associated", _arr(i32, MAX_EXTRA))]


class FrameInfo(C.Structure):
    _fields_ = [("type", i32), ("encoding", i32), ("do_ycbcr", i32), ("upsampling", i32), ("group_dim", i32), ("xqm", i32),
                ("bqm", i32), ("lf_level", i32), ("flags", C.c_uint64), ("jpeg_up_y", _arr(i32, 3)), ("jpeg_up_x", _arr(i32, 3)),
                ("ec_upsampling", _arr(i32, MAX_EXTRA)), ("num_passes", i32), ("pass_shift", _arr(i32, 11)),
                ("x0", i32), ("y0", i32), ("width", i32), ("height", i32), ("padded_width", i32), ("padded_height", i32),
                ("blend_mode", i32), ("blend_alpha", i32), ("blend_clamp", i32), ("blend_source", i32),
                ("ec_blend_mode", _arr(i32, MAX_EXTRA)), ("ec_blend_alpha", _arr(i32, MAX_EXTRA)),
                ("ec_blend_clamp", _arr(i32, MAX_EXTRA)), ("ec_blend_source", _arr(i32, MAX_EXTRA)),
                ("duration", C.c_uint32), ("is_last", i32), ("save_as_reference", i32), ("save_before_ct", i32),
                ("gab", i32), ("epf_iters", i32), ("gab1", _arr(f32, 3)), ("gab2", _arr(f32, 3)), ("epf_sharp_lut", _arr(f32, 8)),
                ("epf_channel_scale", _arr(f32, 3)), ("epf_pass0_sigma", f32), ("epf_pass2_sigma", f32),
                ("epf_border_sad_mul", f32), ("epf_sigma_modular", f32),
                ("num_groups", i32), ("num_lf_groups", i32), ("group_cols", i32), ("lf_group_cols", i32),
                ("num_patches", i32), ("has_splines", i32), ("has_noise", i32), ("noise", _arr(f32, 8)),
                ("lf_dequant", _arr(f32, 3)), ("scaled_dequant", _arr(f32, 3)), ("global_scale", i32), ("quant_lf", i32),
                ("colour_factor", i32), ("x_factor_lf", i32), ("b_factor_lf", i32), ("base_corr_x", f32), ("base_corr_b", f32),
                ("quant_all_default", i32), ("num_hf_presets", i32), ("num_modular_channels", i32)]


class Chan(C.Structure):
    _fields_ = [("w", i32), ("h", i32), ("hshift", i32), ("vshift", i32), ("data", C.POINTER(i32))]


class SqueezeStep(C.Structure):
    _fields_ = [("horizontal", i32), ("in_place", i32), ("begin_c", i32), ("num_c", i32)]


SQUEEZE_CB = C.CFUNCTYPE(i32, C.c_void_p, C.POINTER(Chan), i32, C.POINTER(SqueezeStep), i32, C.POINTER(Chan), i32)
RCT_CB = C.CFUNCTYPE(i32, C.c_void_p, C.POINTER(i32), C.POINTER(i32), C.POINTER(i32), C.c_int64, i32)


class Hooks(C.Structure):
    _fields_ = [("user", C.c_void_p), ("squeeze", SQUEEZE_CB), ("rct", RCT_CB)]


class LFGroupView(C.Structure):
    _fields_ = [("cells_h", i32), ("cells_w", i32), ("extra_precision", i32), ("has_lf_quant", i32),
                ("lf_quant", C.POINTER(i32) * 3), ("lf_h", _arr(i32, 3)), ("lf_w", _arr(i32, 3)), ("n_blocks", i32),
                ("dct_select", C.POINTER(C.c_uint8)), ("hf_mul", C.POINTER(i32)), ("sharpness", C.POINTER(i32)),
                ("x_from_y", C.POINTER(i32)), ("b_from_y", C.POINTER(i32)), ("block_yx", C.POINTER(i32))]


class CoeffView(C.Structure):
    _fields_ = [("q", C.POINTER(i32) * 3), ("h", _arr(i32, 3)), ("w", _arr(i32, 3))]


class QuantView(C.Structure):
    _fields_ = [("mode", i32), ("denominator", f32), ("n_dct", i32), ("n_par", i32), ("n_p44", i32),
                ("dct", C.POINTER(f32)), ("par", C.POINTER(f32)), ("p44", C.POINTER(f32))]


class PatchView(C.Structure):
    _fields_ = [("ref", i32), ("x0", i32), ("y0", i32), ("w", i32), ("h", i32), ("n_positions", i32), ("n_blend", i32),
                ("positions", C.POINTER(i32)), ("blend", C.POINTER(i32))]


class SplineView(C.Structure):
    _fields_ = [("quant_adjust", i32), ("n_control", i32), ("control", C.POINTER(i32)), ("coeff", C.POINTER(i32))]


class FrontendError(RuntimeError):
    def __init__(self, status, msg):
        super().__init__("jxf status %d: %s" % (status, msg))
        self.status = status


SIGNATURES = {
    "jxf_open": (C.c_void_p, [C.c_char_p, C.c_size_t, C.c_char_p, C.c_size_t]),
    "jxf_close": (None, [C.c_void_p]),
    "jxf_last_error": (C.c_char_p, [C.c_void_p]),
    "jxf_get_image_info": (i32, [C.c_void_p, C.POINTER(ImageInfo)]),
    "jxf_get_up_weights": (i32, [C.c_void_p, i32, C.POINTER(f32), i32]),
    "jxf_next_frame": (i32, [C.c_void_p, C.POINTER(Hooks)]),
    "jxf_get_frame_info": (i32, [C.c_void_p, C.POINTER(FrameInfo)]),
    "jxf_get_lfgroup": (i32, [C.c_void_p, i32, C.POINTER(LFGroupView)]),
    "jxf_get_coeffs": (i32, [C.c_void_p, i32, i32, C.POINTER(CoeffView)]),
    "jxf_get_quant_params": (i32, [C.c_void_p, i32, C.POINTER(QuantView)]),
    "jxf_get_patch": (i32, [C.c_void_p, i32, C.POINTER(PatchView)]),
    "jxf_num_splines": (i32, [C.c_void_p]),
    "jxf_get_spline": (i32, [C.c_void_p, i32, C.POINTER(SplineView)]),
    "jxf_get_modular_channel": (i32, [C.c_void_p, i32, C.POINTER(Chan)]),
}

_lib = None


def build(force=False):
    """g++ build of the front-end (jxlatte_amd/frontend/Makefile)."""
    src = os.path.join(HERE, "frontend")
    newest = max(os.path.getmtime(os.path.join(src, f)) for f in os.listdir(src) if f.endswith((".cc", ".h")) or f == "Makefile")
    if force or not os.path.exists(SO_PATH) or os.path.getmtime(SO_PATH) < newest:
        subprocess.check_call(["make", "-C", src, "-s"])
    return SO_PATH


def load():
    global _lib
    if _lib is None:
        build()  # mtime check against the sources: a stale front-end is never loaded silently
        L = C.CDLL(SO_PATH)
        for name, (res, args) in SIGNATURES.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def _np(ptr, shape, dtype=np.int32):
    n = int(np.prod(shape))
    if n == 0 or not ptr:
        return np.zeros(shape, dtype)
    return np.ctypeslib.as_array(ptr, shape=(n,)).reshape(shape).copy()


class Frontend:
    """one open codestream; next_frame() advances, the accessors return numpy copies of the boundary tensors"""

    def __init__(self, data):
        self.lib = load()
        err = C.create_string_buffer(512)
        self._data = bytes(data)
        self.h = self.lib.jxf_open(self._data, len(self._data), err, len(err))
        if not self.h:
            raise FrontendError(-2, err.value.decode("utf-8", "replace"))
        self.image = ImageInfo()
        self._check(self.lib.jxf_get_image_info(self.h, C.byref(self.image)))
        self.frame = None
        self._hooks = None

    def close(self):
        if self.h:
            self.lib.jxf_close(self.h)
            self.h = None

    def __del__(self):
        self.close()

    def _check(self, st):
        if st < 0:
            raise FrontendError(st, self.lib.jxf_last_error(self.h).decode("utf-8", "replace"))
        return st

    def up_weights(self, k_index):
        out = (f32 * 210)()
        n = self._check(self.lib.jxf_get_up_weights(self.h, k_index, out, 210))
        return np.array(out[:n], np.float32)

    def next_frame(self, squeeze=None, rct=None):
        """squeeze(in_channels, steps, out_shapes) -> list of out arrays; rct(v0, v1, v2, rct_type) -> (o0, o1, o2).
        Both operate on numpy int32 arrays; results are copied back into the front-end's buffers."""
        def sq_cb(_user, cin, n_in, steps, n_steps, cout, n_out):
            try:
                ins = [_np(cin[i].data, (cin[i].h, cin[i].w)) for i in range(n_in)]
                st = [(steps[i].horizontal, steps[i].in_place, steps[i].begin_c, steps[i].num_c) for i in range(n_steps)]
                shapes = [(cout[i].h, cout[i].w) for i in range(n_out)]
                outs = squeeze(ins, st, shapes)
                for i in range(n_out):
                    a = np.ascontiguousarray(outs[i], np.int32)
                    assert a.shape == shapes[i], (a.shape, shapes[i])
                    if a.size:
                        C.memmove(cout[i].data, a.ctypes.data, a.nbytes)
                return 0
            except Exception as e:  # noqa: BLE001 - surfaced through the status code
                self._hook_error = e
                return -4

        def rct_cb(_user, v0, v1, v2, n, rct_type):
            try:
                planes = [_np(v, (n,)) for v in (v0, v1, v2)]
                outs = rct(planes[0], planes[1], planes[2], rct_type)
                for dst, a in zip((v0, v1, v2), outs):
                    a = np.ascontiguousarray(a, np.int32)
                    if a.size:
                        C.memmove(dst, a.ctypes.data, a.nbytes)
                return 0
            except Exception as e:  # noqa: BLE001
                self._hook_error = e
                return -4

        self._hook_error = None
        hooks = Hooks()
        hooks.squeeze = SQUEEZE_CB(sq_cb) if squeeze else SQUEEZE_CB()
        hooks.rct = RCT_CB(rct_cb) if rct else RCT_CB()
        self._hooks = hooks
        st = self.lib.jxf_next_frame(self.h, C.byref(hooks))
        if st < 0 and self._hook_error is not None:
            raise self._hook_error
        self._check(st)
        if st == 1:
            self.frame = None
            return None
        self.frame = FrameInfo()
        self._check(self.lib.jxf_get_frame_info(self.h, C.byref(self.frame)))
        return self.frame

    def lfgroup(self, idx):
        v = LFGroupView()
        self._check(self.lib.jxf_get_lfgroup(self.h, idx, C.byref(v)))
        ch, cw = v.cells_h, v.cells_w
        t = ((ch + 7) // 8, (cw + 7) // 8)
        d = dict(cells_h=ch, cells_w=cw, extra_precision=v.extra_precision, n_blocks=v.n_blocks,
                 dct_select=_np(v.dct_select, (ch, cw), np.uint8), hf_mul=_np(v.hf_mul, (ch, cw)),
                 sharpness=_np(v.sharpness, (ch, cw)), x_from_y=_np(v.x_from_y, t), b_from_y=_np(v.b_from_y, t),
                 block_yx=_np(v.block_yx, (v.n_blocks, 2)), lf_quant=None)
        if v.has_lf_quant:
            d["lf_quant"] = [_np(v.lf_quant[i], (v.lf_h[i], v.lf_w[i])) for i in range(3)]
        return d

    def coeffs(self, pass_, group):
        v = CoeffView()
        self._check(self.lib.jxf_get_coeffs(self.h, pass_, group, C.byref(v)))
        return [_np(v.q[c], (v.h[c], v.w[c])) for c in range(3)]

    def quant_params(self, index):
        v = QuantView()
        self._check(self.lib.jxf_get_quant_params(self.h, index, C.byref(v)))

        def arr(p, n):
            return _np(p, (3, n), np.float32) if n else None
        return dict(mode=v.mode, denominator=v.denominator, dct=arr(v.dct, v.n_dct), par=arr(v.par, v.n_par), p44=arr(v.p44, v.n_p44))

    def patch(self, index):
        v = PatchView()
        self._check(self.lib.jxf_get_patch(self.h, index, C.byref(v)))
        return dict(ref=v.ref, x0=v.x0, y0=v.y0, w=v.w, h=v.h, positions=_np(v.positions, (v.n_positions, 2)),
                    blend=_np(v.blend, (v.n_positions, v.n_blend, 3)))

    def splines(self):
        out = []
        for i in range(self.lib.jxf_num_splines(self.h)):
            v = SplineView()
            self._check(self.lib.jxf_get_spline(self.h, i, C.byref(v)))
            out.append(dict(quant_adjust=v.quant_adjust, control=_np(v.control, (v.n_control, 2)), coeff=_np(v.coeff, (4, 32))))
        return out

    def modular_channel(self, index):
        c = Chan()
        self._check(self.lib.jxf_get_modular_channel(self.h, index, C.byref(c)))
        return _np(c.data, (c.h, c.w)), (c.hshift, c.vshift)

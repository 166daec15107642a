"""ctypes mirror of include/jxlatte_amd.h (structures, constants, transform-type table).

Shared by the product binding (jxlatte_amd._lib) and, as plain type definitions, by the
oracle's test wrapper (oracle/pyoracle.py). No compute lives here.
"""
import ctypes as C

import numpy as np

JXL_OK = 0
JXL_ERR_INVALID_ARGUMENT = -1
JXL_ERR_INVALID_BITSTREAM = -2
JXL_ERR_UNSUPPORTED = -3
JXL_ERR_DEVICE = -4
JXL_ERR_OOM = -5
JXL_ERR_STATE = -6

TRANSFER_NONE, TRANSFER_PQ, TRANSFER_SRGB, TRANSFER_PQ_EXACT = 0, 1, 2, 3
OUT_F32, OUT_U16, OUT_U8, OUT_RGB8, OUT_RGB16 = 0, 1, 2, 3, 4
BLEND_REPLACE, BLEND_ADD, BLEND_BLEND, BLEND_MULADD, BLEND_MULT = 0, 1, 2, 3, 4
BLEND_FLAG_IS_ALPHA, BLEND_FLAG_HAS_EXTRA, BLEND_FLAG_CLAMP, BLEND_FLAG_PREMULT = 1, 2, 4, 8
STAGE_IDCT, STAGE_GAB, STAGE_EPF, STAGE_XYB, STAGE_OUT = 1, 2, 4, 8, 16
STAGE_ALL = 31

# (type, parameterIndex, orderID, method, pixelHeight, pixelWidth): TransformType.java:10-36
# (kept in sync with include/jxl_transform_types.h; tests/test_abi.py checks it against the .so)
METHOD_DCT, METHOD_DCT2, METHOD_DCT4, METHOD_HORNUSS, METHOD_DCT8_4, METHOD_DCT4_8, METHOD_AFV = range(7)
TRANSFORM_TYPES = [
    ("DCT8", 0, 0, 0, METHOD_DCT, 8, 8), ("HORNUSS", 1, 1, 1, METHOD_HORNUSS, 8, 8),
    ("DCT2", 2, 2, 1, METHOD_DCT2, 8, 8), ("DCT4", 3, 3, 1, METHOD_DCT4, 8, 8),
    ("DCT16", 4, 4, 2, METHOD_DCT, 16, 16), ("DCT32", 5, 5, 3, METHOD_DCT, 32, 32),
    ("DCT16_8", 6, 6, 4, METHOD_DCT, 16, 8), ("DCT8_16", 7, 6, 4, METHOD_DCT, 8, 16),
    ("DCT32_8", 8, 7, 5, METHOD_DCT, 32, 8), ("DCT8_32", 9, 7, 5, METHOD_DCT, 8, 32),
    ("DCT32_16", 10, 8, 6, METHOD_DCT, 32, 16), ("DCT16_32", 11, 8, 6, METHOD_DCT, 16, 32),
    ("DCT4_8", 12, 9, 1, METHOD_DCT4_8, 8, 8), ("DCT8_4", 13, 9, 1, METHOD_DCT8_4, 8, 8),
    ("AFV0", 14, 10, 1, METHOD_AFV, 8, 8), ("AFV1", 15, 10, 1, METHOD_AFV, 8, 8),
    ("AFV2", 16, 10, 1, METHOD_AFV, 8, 8), ("AFV3", 17, 10, 1, METHOD_AFV, 8, 8),
    ("DCT64", 18, 11, 7, METHOD_DCT, 64, 64), ("DCT64_32", 19, 12, 8, METHOD_DCT, 64, 32),
    ("DCT32_64", 20, 12, 8, METHOD_DCT, 32, 64), ("DCT128", 21, 13, 9, METHOD_DCT, 128, 128),
    ("DCT128_64", 22, 14, 10, METHOD_DCT, 128, 64), ("DCT64_128", 23, 14, 10, METHOD_DCT, 64, 128),
    ("DCT256", 24, 15, 11, METHOD_DCT, 256, 256), ("DCT256_128", 25, 16, 12, METHOD_DCT, 256, 128),
    ("DCT128_256", 26, 16, 12, METHOD_DCT, 128, 256),
]
TT_NAME = {t[1]: t[0] for t in TRANSFORM_TYPES}
TT_BY_NAME = {t[0]: t[1] for t in TRANSFORM_TYPES}


def tt_pixel_size(t):
    return TRANSFORM_TYPES[t][5], TRANSFORM_TYPES[t][6]


def tt_param_index(t):
    return TRANSFORM_TYPES[t][2]


def tt_matrix_size(param_index):
    """matrixHeight x matrixWidth of the (non-vertical) type of a parameter index
    (TransformType.getByParameterIndex, TransformType.java:70-73)."""
    for _, _, p, _, _, ph, pw in TRANSFORM_TYPES:
        if p == param_index and not ph > pw:
            return min(ph, pw), max(ph, pw)
    raise KeyError(param_index)


def iptr(a):
    assert a.dtype == np.int32
    return ptr(a, C.c_int32)


f3 = C.c_float * 3
f8 = C.c_float * 8
f9 = C.c_float * 9


class VarDCTParams(C.Structure):
    """struct jxl_vardct_params"""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32), ("stages", C.c_uint32),
        ("scale_factor", f3), ("quant_bias", f3), ("quant_bias_numerator", C.c_float),
        ("base_corr_x", C.c_float), ("base_corr_b", C.c_float), ("color_factor", C.c_int32),
        ("gab", C.c_int32), ("gab_w1", f3), ("gab_w2", f3),
        ("epf_iters", C.c_int32), ("global_scale_f", C.c_float), ("epf_sharp_lut", f8),
        ("epf_channel_scale", f3), ("epf_pass0_sigma_scale", C.c_float),
        ("epf_pass2_sigma_scale", C.c_float), ("epf_border_sad_mul", C.c_float),
        ("xyb", C.c_int32), ("opsin_matrix", f9), ("opsin_bias", f3), ("cbrt_opsin_bias", f3),
        ("intensity_target", C.c_float), ("transfer", C.c_int32), ("out_format", C.c_int32),
        ("jpeg_upsampling_y", C.c_int32 * 3), ("jpeg_upsampling_x", C.c_int32 * 3),
    ]


class LFGroupDesc(C.Structure):
    """struct jxl_lfgroup_desc"""
    _fields_ = [
        ("lfg_y", C.c_int32), ("lfg_x", C.c_int32), ("cells_h", C.c_int32), ("cells_w", C.c_int32),
        ("dct_select", C.POINTER(C.c_uint8)), ("hf_mul", C.POINTER(C.c_int32)),
        ("sharpness", C.POINTER(C.c_int32)), ("x_from_y", C.POINTER(C.c_int32)),
        ("b_from_y", C.POINTER(C.c_int32)), ("block_yx", C.POINTER(C.c_int32)),
        ("n_blocks", C.c_int32), ("lf", C.POINTER(C.c_float) * 3),
    ]


class LFQuantDesc(C.Structure):
    """struct jxl_lfquant_desc (row f1: integer LF image of one LF group)"""
    _fields_ = [
        ("lfg_y", C.c_int32), ("lfg_x", C.c_int32), ("cells_h", C.c_int32), ("cells_w", C.c_int32),
        ("lf_quant", C.POINTER(C.c_int32) * 3), ("extra_precision", C.c_int32), ("scaled_dequant", f3),
        ("x_factor_lf", C.c_int32), ("b_factor_lf", C.c_int32), ("adaptive_smoothing", C.c_int32),
    ]


def make_lfquant_desc(lf_quant, scaled_dequant, extra_precision=0, x_factor_lf=128, b_factor_lf=128, adaptive_smoothing=True,
                      lfg_y=0, lfg_x=0, cells=None):
    """lf_quant: int32 array [3][H][W] in X,Y,B order, or (chroma-subsampled frames) a list of three contiguous int32 planes
    of different sizes together with cells=(cells_h, cells_w) of the LF group. Kept alive by the caller."""
    planes = [lf_quant[c] for c in range(3)]
    for a in planes:
        assert a.dtype == np.int32 and a.ndim == 2 and a.flags["C_CONTIGUOUS"]
    d = LFQuantDesc()
    d.lfg_y, d.lfg_x = lfg_y, lfg_x
    d.cells_h, d.cells_w = cells if cells is not None else max(a.shape for a in planes)
    for c in range(3):
        d.lf_quant[c] = iptr(planes[c])
        d.scaled_dequant[c] = scaled_dequant[c]
    d.extra_precision, d.x_factor_lf, d.b_factor_lf = extra_precision, x_factor_lf, b_factor_lf
    d.adaptive_smoothing = 1 if adaptive_smoothing else 0
    return d


class BlendRect(C.Structure):
    """struct jxl_blend_rect"""
    _fields_ = [("h", C.c_int32), ("w", C.c_int32), ("canvas_y", C.c_int32), ("canvas_x", C.c_int32),
                ("frame_y", C.c_int32), ("frame_x", C.c_int32), ("ref_y", C.c_int32), ("ref_x", C.c_int32)]


class PackParams(C.Structure):
    """struct jxl_pack_params"""
    _fields_ = [("height", C.c_int32), ("width", C.c_int32), ("n_color", C.c_int32), ("has_alpha", C.c_int32),
                ("premultiplied", C.c_int32), ("bit_depth", C.c_int32), ("big_endian", C.c_int32),
                ("is_int", C.c_int32 * 4), ("tagged_depth", C.c_int32 * 4)]


class SqueezeParam(C.Structure):
    """struct jxl_squeeze_param (SqueezeParam.java)"""
    _fields_ = [("horizontal", C.c_int32), ("in_place", C.c_int32), ("begin_c", C.c_int32), ("num_c", C.c_int32)]

    def as_tuple(self):
        return (self.horizontal, self.in_place, self.begin_c, self.num_c)


class Channel(C.Structure):
    """struct jxl_channel"""
    _fields_ = [("width", C.c_int32), ("height", C.c_int32), ("data", C.POINTER(C.c_int32))]


def ptr(a, ctype):
    """pointer to a C-contiguous numpy array's data (array must stay alive)."""
    assert a.flags["C_CONTIGUOUS"], "array must be C-contiguous"
    return a.ctypes.data_as(C.POINTER(ctype))


def fptr(a):
    assert a.dtype == np.float32
    return ptr(a, C.c_float)




def make_lfgroup_desc(g):
    """g: dict with numpy arrays (see synth.make_vardct_frame). Returns (desc, keepalive)."""
    d = LFGroupDesc()
    d.lfg_y, d.lfg_x = int(g["lfg_y"]), int(g["lfg_x"])
    d.cells_h, d.cells_w = g["dct_select"].shape
    d.dct_select = ptr(g["dct_select"], C.c_uint8)
    d.hf_mul = iptr(g["hf_mul"])
    d.sharpness = iptr(g["sharpness"])
    d.x_from_y = iptr(g["x_from_y"])
    d.b_from_y = iptr(g["b_from_y"])
    d.block_yx = iptr(g["block_yx"])
    d.n_blocks = g["block_yx"].shape[0]
    if g.get("lf") is not None:
        for c in range(3):
            d.lf[c] = fptr(g["lf"][c])
    return d


def make_channels(arrs):
    """list of 2-D int32 arrays -> (Channel array, keepalive)."""
    arr = (Channel * max(1, len(arrs)))()
    for i, a in enumerate(arrs):
        assert a.dtype == np.int32 and a.ndim == 2 and a.flags["C_CONTIGUOUS"]
        arr[i].height, arr[i].width = a.shape
        arr[i].data = iptr(a) if a.size else C.POINTER(C.c_int32)()
    return arr


def make_squeeze_params(sp):
    arr = (SqueezeParam * max(1, len(sp)))()
    for i, (h, ip, b, n) in enumerate(sp):
        arr[i].horizontal, arr[i].in_place, arr[i].begin_c, arr[i].num_c = int(h), int(ip), int(b), int(n)
    return arr

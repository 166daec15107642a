"""ctypes binding of libjxlatte_amd.so (the C-ABI of include/jxlatte_amd.h).

There is no CPU fallback: if the HIP library is missing, or no GPU is present when a context
is created, the product path fails loudly (LibraryMissing / JxlError).
"""
import ctypes as C
import os

from . import abi

HERE = os.path.dirname(os.path.abspath(__file__))
SO_PATH = os.environ.get("JXL_AMD_LIB") or os.path.join(HERE, "libjxlatte_amd.so")  # override: A/B runs of two builds


class LibraryMissing(RuntimeError):
    pass


class JxlError(RuntimeError):
    """Base error; `.status` is the jxl_status code."""

    def __init__(self, status, msg):
        super().__init__("jxl status %d: %s" % (status, msg))
        self.status = status


class InvalidBitstreamException(JxlError, IOError):
    """mirrors com.traneptora.jxlatte.io.InvalidBitstreamException (extends IOException)"""


class UnsupportedOperationException(JxlError):
    """mirrors java.lang.UnsupportedOperationException (PassGroup.java:326-327)"""


class IllegalArgumentException(JxlError, ValueError):
    pass


class IllegalStateException(JxlError):
    pass


_ERR = {
    abi.JXL_ERR_INVALID_ARGUMENT: IllegalArgumentException,
    abi.JXL_ERR_INVALID_BITSTREAM: InvalidBitstreamException,
    abi.JXL_ERR_UNSUPPORTED: UnsupportedOperationException,
    abi.JXL_ERR_STATE: IllegalStateException,
}

i32, i64, f32, vp = C.c_int32, C.c_int64, C.c_float, C.c_void_p
pf, pi, pu8 = C.POINTER(C.c_float), C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
pf3 = C.POINTER(pf)  # const float* const [3]
pi3 = C.POINTER(pi)
pv3 = C.POINTER(vp)

# every symbol include/jxlatte_amd.h declares: name -> (restype, argtypes)
SIGNATURES = {
    "jxl_ctx_create": (i32, [i32, C.POINTER(vp)]),
    "jxl_ctx_destroy": (None, [vp]),
    "jxl_last_error": (C.c_char_p, [vp]),
    "jxl_version": (C.c_char_p, []),
    "jxl_ctx_synchronize": (i32, [vp]),
    "jxl_ctx_stream": (vp, [vp]),
    "jxl_ctx_set_stream": (i32, [vp, vp]),
    "jxl_vardct_begin_frame": (i32, [vp, C.POINTER(abi.VarDCTParams)]),
    "jxl_vardct_set_weights": (i32, [vp, pf, C.c_size_t, pi]),
    "jxl_vardct_set_lfgroup": (i32, [vp, C.POINTER(abi.LFGroupDesc)]),
    "jxl_vardct_set_lfgroup_lfquant": (i32, [vp, C.POINTER(abi.LFQuantDesc)]),
    "jxl_vardct_put_group": (i32, [vp, i32, i32, pi3, pi]),
    "jxl_vardct_put_group_i16": (i32, [vp, i32, i32, C.POINTER(C.POINTER(C.c_int16)), pi]),
    "jxl_vardct_map_coeffs_i16": (i32, [vp, C.POINTER(C.POINTER(C.c_int16)), pi]),
    "jxl_vardct_coeff_plane_rows": (i32, [vp, pi]),
    "jxl_vardct_geometry": (i32, [vp, pi]),
    "jxl_vardct_output_geometry": (i32, [vp, pi]),
    "jxl_vardct_group_size": (i32, [vp, i32, pi, pi]),
    "jxl_vardct_commit_coeffs_i16": (i32, [vp]),
    "jxl_vardct_map_coeffs_i16_ex": (i32, [vp, C.POINTER(C.POINTER(C.c_int16)), pi, i32]),
    "jxl_vardct_commit_coeffs_i16_groups": (i32, [vp, C.POINTER(C.c_uint8), i32]),
    "jxl_host_alloc": (vp, [C.c_size_t]),
    "jxl_host_free": (None, [vp]),
    "jxl_vardct_prepare": (i32, [vp]),
    "jxl_vardct_run": (i32, [vp]),
    "jxl_vardct_run_batch": (i32, [C.POINTER(C.c_void_p), i32]),
    "jxl_vardct_finish_frame": (i32, [vp, pv3, i64]),
    "jxl_vardct_read_output": (i32, [vp, pv3, i64]),
    "jxl_vardct_read_output_begin": (i32, [vp, pv3, i64]),
    "jxl_vardct_read_output_wait": (i32, [vp]),
    "jxl_vardct_copy_output_device": (i32, [vp, vp]),
    "jxl_planes_from_frame": (i32, [vp, i32, i32]),
    "jxl_planes_upsample": (i32, [vp, i32, pf]),
    "jxl_planes_noise": (i32, [vp, i32, C.c_uint64, pf, f32, f32]),
    "jxl_planes_xyb": (i32, [vp, pf, pf, pf, f32]),
    "jxl_planes_ycbcr": (i32, [vp]),
    "jxl_planes_shape": (i32, [vp, pi, pi]),
    "jxl_planes_download": (i32, [vp, pf3]),
    "jxl_planes_upload": (i32, [vp, pf3, i32, i32]),
    "jxl_vardct_out_elem_size": (i32, [vp]),
    "jxl_vardct_last_launch_count": (i32, [vp]),
    "jxl_vardct_last_stage_ms": (i32, [vp, i32, pf]),
    "jxl_vardct_enable_stage_timing": (i32, [vp, i32]),
    "jxl_stage_idct2d": (i32, [vp, pf, pf, i32, i32, i32]),
    "jxl_stage_fdct2d": (i32, [vp, pf, pf, i32, i32]),
    "jxl_stage_gab": (i32, [vp, pf3, pf3, i32, i32, pf, pf]),
    "jxl_stage_epf": (i32, [vp, pf3, pf3, i32, i32, i32, pf, f32, pf, f32, f32, f32]),
    "jxl_stage_epf_sigma": (i32, [vp, pi, pi, i32, i32, f32, pf, pf]),
    "jxl_stage_lf_dequant": (i32, [vp, C.POINTER(abi.LFQuantDesc), f32, f32, i32, pf3]),
    "jxl_stage_xyb": (i32, [vp, pf3, i64, pf, pf, pf, f32]),
    "jxl_stage_ycbcr": (i32, [vp, pf3, i64]),
    "jxl_stage_transfer": (i32, [vp, pf, i64, i32, i32, pf, pi]),
    "jxl_stage_inv_hsqueeze": (i32, [vp, pi, i32, pi, i32, i32, pi]),
    "jxl_stage_inv_vsqueeze": (i32, [vp, pi, i32, pi, i32, i32, pi]),
    "jxl_stage_rct": (i32, [vp, pi3, i64, i32]),
    "jxl_stage_modular_to_float": (i32, [vp, pi, pi, i64, f32, pf]),
    "jxl_stage_chroma_upsample": (i32, [vp, pf, i32, i32, i32, i32, pf]),
    "jxl_upsampling_weights": (i32, [i32, pf, pf]),
    "jxl_stage_upsample": (i32, [vp, pf, i32, i32, i32, pf, pf]),
    "jxl_stage_noise_init": (i32, [vp, i32, i32, i32, C.c_uint64, i32, pf3]),
    "jxl_stage_noise_add": (i32, [vp, pf3, pf3, i64, pf, f32, f32]),
    "jxl_stage_blend": (i32, [vp, i32, C.c_uint32, i32, vp, i32, i32, vp, i32, i32, vp, i32, i32, pf, pf,
                              C.POINTER(abi.BlendRect)]),
    "jxl_stage_orient": (i32, [vp, vp, i32, i32, i32, vp]),
    "jxl_stage_pack": (i32, [vp, pv3, C.POINTER(abi.PackParams), vp]),
    "jxl_modular_default_squeeze_params": (i32, [pi, pi, i32, i32, C.POINTER(abi.SqueezeParam), i32]),
    "jxl_modular_squeezed_shapes": (i32, [pi, pi, i32, C.POINTER(abi.SqueezeParam), i32, pi, pi, i32]),
    "jxl_modular_begin": (i32, [vp, C.POINTER(abi.Channel), i32, C.POINTER(abi.SqueezeParam), i32, i32, i32]),
    "jxl_modular_run": (i32, [vp]),
    "jxl_modular_out_count": (i32, [vp]),
    "jxl_modular_out_shape": (i32, [vp, i32, pi, pi]),
    "jxl_modular_read_channel": (i32, [vp, i32, pi]),
    "jxl_modular_apply": (i32, [vp, C.POINTER(abi.Channel), i32, C.POINTER(abi.SqueezeParam), i32, i32, i32,
                                C.POINTER(abi.Channel), i32]),
    "jxl_modular_last_launch_count": (i32, [vp]),
    "jxl_modular_redo_count": (i32, [vp]),
}

_lib = None


def load():
    """dlopen the HIP library and bind every declared symbol (no device needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise LibraryMissing(
            "%s is missing: build it with `python -m jxlatte_amd.build` (hipcc, gfx950). "
            "jxlatte_amd has no CPU fallback." % SO_PATH)
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        try:
            fn = getattr(lib, name)  # AttributeError if the .so does not export it
        except AttributeError:
            # an A/B run against an OLDER build (JXL_AMD_LIB=..., JXL_AMD_LIB_OLD=1: tools/ab_idct.sh): entries it lacks stay unbound
            if os.environ.get("JXL_AMD_LIB") and os.environ.get("JXL_AMD_LIB_OLD"):
                continue
            raise
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(ctx_handle, status):
    if status == 0:
        return
    msg = load().jxl_last_error(ctx_handle)
    msg = msg.decode("utf-8", "replace") if msg else ""
    raise _ERR.get(status, JxlError)(status, msg)


class Context:
    """jxl_ctx: one HIP device + one stream + device arena. Like a JXLDecoder instance it is
    single-threaded; distinct contexts are independent."""

    def __init__(self, device=0):
        self.lib = load()
        h = vp()
        st = self.lib.jxl_ctx_create(device, C.byref(h))
        if st != 0:
            msg = self.lib.jxl_last_error(None)
            raise _ERR.get(st, JxlError)(st, msg.decode() if msg else "")
        self.h = h
        self.device = device

    def close(self):
        if getattr(self, "h", None):
            self.lib.jxl_ctx_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def call(self, name, *args):
        check(self.h, getattr(self.lib, name)(self.h, *args))

    def synchronize(self):
        self.call("jxl_ctx_synchronize")

    @property
    def stream(self):
        return self.lib.jxl_ctx_stream(self.h)

"""The streaming boundary leg's pieces: the native host-thread harness of bench.py (tools/native/stream_bench.cpp) and the
kernel-driven transfers behind jxl_vardct_read_output* / commit_coeffs_i16 (r5)."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from jxlatte_amd import _lib, abi, host, synth  # noqa: E402
from jxlatte_amd import build as hip_build  # noqa: E402


def test_native_stream_harness_builds_and_matches_its_binding():
    import bench
    so = hip_build.build_stream_bench()
    drv = C.CDLL(so)
    drv.jxl_stream_bench_args_size.restype = C.c_size_t
    assert drv.jxl_stream_bench_args_size() == C.sizeof(bench._StreamBenchArgs)
    assert hasattr(drv, "jxl_stream_bench")


def _frame(w, h, seed, out_format):
    fr = synth.make_vardct_frame(w, h, seed=seed, aligned=False)
    p = abi.VarDCTParams.from_buffer_copy(fr["params"])
    p.transfer, p.out_format, p.stages = abi.TRANSFER_SRGB, out_format, 31
    return fr, p


def _sync_output(ctx, fr, p):
    f = host.Frame(ctx, p, fr["weights"], fr["woffs"])
    for g in fr["lfgroups"]:
        f.setLFGroup(g)
    planes = f.mapCoeffsI16()
    for c in range(3):
        planes[c][...] = fr["coeff"][c]
    f.commitCoeffsI16()
    return f, f.decodeFrame()  # pageable destination: the runtime's copy


@pytest.mark.gpu
@pytest.mark.parametrize("size,fmt", [((520, 264), "RGB8"), ((264, 520), "RGB16"), ((72, 40), "U8"), ((1032, 520), "U16")])
def test_page_locked_output_is_written_by_a_kernel_and_equals_the_runtime_copy(size, fmt):
    """a page-locked destination takes the k_copy16 path (copy_zero), a pageable one hipMemcpyAsync: same bytes, for the
    interleaved and the planar integer sinks, with read_output and with read_output_begin / _wait"""
    lib = _lib.load()
    fr, p = _frame(size[0], size[1], 90, getattr(abi, "OUT_" + fmt))
    with _lib.Context(0) as ctx:
        f, exp = _sync_output(ctx, fr, p)
        pin = host.PinnedArray(lib, exp.shape, exp.dtype)
        try:
            pin.array[...] = 0
            if exp.ndim == 3 and exp.shape[-1] == 3:
                pp = (C.c_void_p * 3)(pin.array.ctypes.data, None, None)
            else:
                pp = (C.c_void_p * 3)(*[pin.array[c].ctypes.data for c in range(3)])
            ctx.call("jxl_vardct_read_output", pp, f.width)
            assert np.array_equal(pin.array, exp)
            pin.array[...] = 0
            ctx.call("jxl_vardct_read_output_begin", pp, f.width)
            ctx.call("jxl_vardct_read_output_wait")
            assert np.array_equal(pin.array, exp)
        finally:
            pin.free()


@pytest.mark.gpu
def test_page_locked_output_shorter_than_the_frame_is_an_error_not_a_fault():
    """r6: copy_zero writes a page-locked destination through its device alias with a kernel; it now looks up the
    extent of the allocation first (hipMemGetAddressRange) and a destination SHORTER than the frame is JXL_ERR_INVALID_ARGUMENT -- not a
    GPU page fault past the end of the registration --, for read_output and read_output_begin alike. The context stays usable."""
    lib = _lib.load()
    fr, p = _frame(520, 264, 91, abi.OUT_F32)
    with _lib.Context(0) as ctx:
        f, exp = _sync_output(ctx, fr, p)
        plane = exp[0].nbytes
        short = host.PinnedArray(lib, (3 * plane - 4096,), np.uint8)  # three planes carved from one allocation; the last one is 4 KB short
        full = host.PinnedArray(lib, exp.shape, exp.dtype)
        try:
            pp = (C.c_void_p * 3)(*[short.array.ctypes.data + c * plane for c in range(3)])
            with pytest.raises(_lib.IllegalArgumentException):
                ctx.call("jxl_vardct_read_output", pp, f.width)
            with pytest.raises(_lib.IllegalArgumentException):
                ctx.call("jxl_vardct_read_output_begin", pp, f.width)
            ctx.synchronize()
            pq = (C.c_void_p * 3)(*[full.array[c].ctypes.data for c in range(3)])
            ctx.call("jxl_vardct_read_output", pq, f.width)
            assert np.array_equal(full.array.view(np.uint32), exp.view(np.uint32))
        finally:
            ctx.synchronize()
            short.free()
            full.free()


@pytest.mark.gpu
@pytest.mark.parametrize("n_ctx,fpc", [(1, 2), (3, 4)])
def test_native_stream_harness_streams_identical_frames(n_ctx, fpc):
    """bench.py's streaming leg at a small size: every context's last frame equals the synchronous path's pixels"""
    import bench
    fr, p = _frame(520, 520, 91, abi.OUT_RGB8)
    with _lib.Context(0) as ctx:
        _, exp = _sync_output(ctx, fr, p)
    r = bench.streaming_leg_native(_lib, host, fr, p, 0, 520 * 520, exp, n_ctx, fpc)
    assert r is not None and "error" not in r, r
    assert r["identical_output"] and r["frames"] == n_ctx * fpc

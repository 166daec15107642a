"""integration/jni/jxlatte_amd_jni.c, kept from rotting without a JDK:

* CPU: the shim parses cleanly (`gcc -fsyntax-only -Wall -Wextra -Werror`) against tests/stubs/jni.h -- our own declaration of the JNI
  types and the function-table entries the shim uses -- and exports one `Java_..._NativeBackend_<name>` symbol per `native` method of
  integration/jni/NativeBackend.java.
* GPU: the shim's entry points are CALLED, through ctypes, over tests/stubs/fake_jni.c (those table entries implemented over plain C
  structs; direct buffers and arrays are ordinary memory): a frame fed and read back through the JNI entries equals the same frame
  through the C-ABI; the output size guards follow jxl_vardct_output_geometry (interleaved RGB8 needs 3 x the plane in ONE buffer and
  no oy / ob; planar needs three); errors arrive as the Java exception classes INTEGRATION.md names.

Neither says anything about a JVM (the stub's table is not the JDK's); it checks the C of the shim and its argument checks."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "integration", "jni", "jxlatte_amd_jni.c")
STUBS = os.path.join(ROOT, "tests", "stubs")
INC = os.path.join(ROOT, "include")
PKG = os.path.join(ROOT, "jxlatte_amd")
PREFIX = "Java_com_traneptora_jxlatte_gpu_NativeBackend_"


def test_shim_parses_against_the_stub_header():
    r = subprocess.run(["gcc", "-std=c11", "-fsyntax-only", "-Wall", "-Wextra", "-Werror", "-I", STUBS, "-I", INC, SHIM],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def _build(tmp):
    so = os.path.join(str(tmp), "libjxl_fakejni.so")
    cmd = ["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-fPIC", "-shared", "-fvisibility=hidden", "-I", STUBS, "-I", INC, SHIM,
           os.path.join(STUBS, "fake_jni.c"), "-o", so, "-L", PKG, "-ljxlatte_amd", "-Wl,-rpath," + PKG]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    return so


def test_every_native_method_has_an_entry_point(tmp_path):
    so = _build(tmp_path)
    syms = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True, check=True).stdout
    exported = set(re.findall(r"\b%s(\w+)" % PREFIX, syms))
    java = open(os.path.join(ROOT, "integration", "jni", "NativeBackend.java")).read()
    natives = set(re.findall(r"\bnative\s+[\w\[\]<>.]+\s+(\w+)\s*\(", java))
    assert natives, "no native methods found in NativeBackend.java"
    assert natives <= exported, "native methods without an entry point in the shim: %s" % sorted(natives - exported)
    assert exported <= natives, "entry points no Java method declares: %s" % sorted(exported - natives)


# ---- GPU: the entry points called over the fake environment -------------------------------------------------------------------------
class FakeJVM:
    def __init__(self, so):
        self.lib = L = C.CDLL(so)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        for name, res, args in (("fj_env_new", vp, []), ("fj_pending_class", C.c_char_p, [vp]), ("fj_pending_message", C.c_char_p, [vp]),
                                ("fj_clear", None, [vp]), ("fj_self", vp, [i64]), ("fj_direct", vp, [vp, i64]), ("fj_ints", vp, [vp, i64]),
                                ("fj_floats", vp, [vp, i64]), ("fj_bytes", vp, [vp, i64]), ("fj_longs", vp, [vp, i64]),
                                ("fj_get_object", vp, [vp, i64]), ("fj_data", vp, [vp]), ("fj_length", i64, [vp])):
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        self.env = L.fj_env_new()
        self.keep = []

    def fn(self, name, res, *argtypes):
        f = getattr(self.lib, PREFIX + name)
        f.restype, f.argtypes = res, [C.c_void_p, C.c_void_p] + list(argtypes)
        return f

    def direct(self, a, nbytes=None):
        """a direct ByteBuffer over a numpy array (or null)"""
        if a is None:
            return None
        self.keep.append(a)
        return self.lib.fj_direct(a.ctypes.data, a.nbytes if nbytes is None else nbytes)

    def ints(self, a):
        a = np.ascontiguousarray(a, np.int32)
        return self.lib.fj_ints(a.ctypes.data, a.size)

    def pending(self):
        c = self.lib.fj_pending_class(self.env)
        return None if c is None else (c.decode(), self.lib.fj_pending_message(self.env).decode())

    def take(self):
        p = self.pending()
        self.lib.fj_clear(self.env)
        return p


def _feed(vm, self_, frame, params):
    """the calls GpuFrameBridge makes for one frame: beginFrame, setWeights, setLFGroup per LF group, putGroup per group"""
    from jxlatte_amd import synth
    i32, vp = C.c_int32, C.c_void_p
    pbuf = np.frombuffer(bytes(params), np.uint8).copy()
    vm.fn("beginFrame", None, vp)(vm.env, self_, vm.direct(pbuf))
    assert vm.pending() is None, vm.pending()
    w = np.ascontiguousarray(frame["weights"], np.float32)
    vm.fn("setWeights", None, vp, vp)(vm.env, self_, vm.direct(w), vm.ints(frame["woffs"]))
    assert vm.pending() is None, vm.pending()
    set_lfg = vm.fn("setLFGroup", None, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp)
    for g in frame["lfgroups"]:
        a = {k: np.ascontiguousarray(g[k]) for k in ("dct_select", "hf_mul", "sharpness", "x_from_y", "b_from_y", "block_yx")}
        lf = [np.ascontiguousarray(p, np.float32) for p in g["lf"]]
        ch, cw = a["dct_select"].shape
        set_lfg(vm.env, self_, int(g["lfg_y"]), int(g["lfg_x"]), ch, cw, vm.direct(a["dct_select"].astype(np.uint8)),
                vm.direct(a["hf_mul"].astype(np.int32)), vm.direct(a["sharpness"].astype(np.int32)), vm.direct(a["x_from_y"].astype(np.int32)),
                vm.direct(a["b_from_y"].astype(np.int32)), vm.direct(a["block_yx"].astype(np.int32)), a["block_yx"].shape[0],
                vm.direct(lf[0]), vm.direct(lf[1]), vm.direct(lf[2]))
        assert vm.pending() is None, vm.pending()
    put = vm.fn("putGroup", None, i32, i32, vp, vp, vp, i32, i32, i32)
    for grp in range(synth.num_groups(frame)):
        q = [np.ascontiguousarray(a, np.int32) for a in synth.group_view(frame, grp)]
        put(vm.env, self_, 0, grp, vm.direct(q[0]), vm.direct(q[1]), vm.direct(q[2]), q[0].shape[1], q[1].shape[1], q[2].shape[1])
        assert vm.pending() is None, vm.pending()


@pytest.mark.gpu
def test_frame_through_the_jni_entries_equals_the_c_abi(ctx, tmp_path):
    from conftest import assert_bits_equal
    from jxlatte_amd import abi, host, synth
    vm = FakeJVM(_build(tmp_path))
    i32, i64, vp = C.c_int32, C.c_int64, C.c_void_p
    frame = synth.make_vardct_frame(320, 200, seed=41, mix="all")
    exp = host.Frame.from_synth(ctx, frame).decodeFrame()
    H, W = exp.shape[1:]

    handle = vm.fn("create", i64, i32)(vm.env, None, 0)
    assert handle and vm.pending() is None
    self_ = vm.lib.fj_self(handle)
    try:
        params = abi.VarDCTParams.from_buffer_copy(frame["params"])
        _feed(vm, self_, frame, params)
        vm.fn("run", None)(vm.env, self_)
        assert vm.pending() is None, vm.pending()
        read = vm.fn("readOutput", None, vp, vp, vp, i64)
        out = [np.zeros((H, W), np.float32) for _ in range(3)]
        read(vm.env, self_, vm.direct(out[0]), vm.direct(out[1]), vm.direct(out[2]), W)
        assert vm.pending() is None, vm.pending()
        assert_bits_equal(np.stack(out), exp, "JNI entries vs C-ABI")

        # planar output: every one of the three buffers is needed, and at the full plane size
        read(vm.env, self_, vm.direct(out[0]), None, vm.direct(out[2]), W)
        assert vm.take()[0] == "java/lang/IllegalArgumentException"
        read(vm.env, self_, vm.direct(out[0]), vm.direct(out[1]), vm.direct(out[2], out[2].nbytes - 4), W)
        assert vm.take()[0] == "java/lang/IllegalArgumentException"
        read(vm.env, self_, vm.direct(out[0]), vm.direct(out[1]), vm.direct(out[2]), W - 1)  # stride shorter than a row
        assert vm.take()[0] == "java/lang/IllegalArgumentException"

        # interleaved RGB8: 3 * W * H bytes in ox alone; oy / ob may be null; a buffer of ONE plane's size is refused
        # before the library writes 3 x that into it
        params.transfer, params.out_format = abi.TRANSFER_SRGB, abi.OUT_RGB8
        fr8 = dict(frame)
        fr8["params"] = bytes(params)
        exp8 = host.Frame.from_synth(ctx, fr8).decodeFrame()
        _feed(vm, self_, frame, params)
        vm.fn("run", None)(vm.env, self_)
        assert vm.pending() is None, vm.pending()
        rgb = np.zeros((H, W, 3), np.uint8)
        read(vm.env, self_, vm.direct(rgb), None, None, W)
        assert vm.pending() is None, vm.pending()
        assert np.array_equal(rgb.reshape(exp8.shape) if rgb.shape != exp8.shape else rgb, exp8)
        small = np.zeros((H, W), np.uint8)
        read(vm.env, self_, vm.direct(small), vm.direct(small.copy()), vm.direct(small.copy()), W)
        cls, msg = vm.take()
        assert cls == "java/lang/IllegalArgumentException" and "too small" in msg

        # error mapping: a transform type outside 0..26 is the reference's InvalidBitstreamException (HFMetadata / TransformType)
        bad = {k: (v if k != "lfgroups" else [dict(g) for g in v]) for k, v in frame.items()}
        sel = np.array(bad["lfgroups"][0]["dct_select"], copy=True)
        sel.flat[0] = 27
        bad["lfgroups"][0]["dct_select"] = sel
        params.transfer, params.out_format = abi.TRANSFER_NONE, abi.OUT_F32
        pbuf = np.frombuffer(bytes(params), np.uint8).copy()
        vm.fn("beginFrame", None, vp)(vm.env, self_, vm.direct(pbuf))
        g = bad["lfgroups"][0]
        ch, cw = g["dct_select"].shape
        vm.fn("setLFGroup", None, i32, i32, i32, i32, vp, vp, vp, vp, vp, vp, i32, vp, vp, vp)(
            vm.env, self_, int(g["lfg_y"]), int(g["lfg_x"]), ch, cw, vm.direct(np.ascontiguousarray(g["dct_select"], np.uint8)),
            vm.direct(np.ascontiguousarray(g["hf_mul"], np.int32)), vm.direct(np.ascontiguousarray(g["sharpness"], np.int32)),
            vm.direct(np.ascontiguousarray(g["x_from_y"], np.int32)), vm.direct(np.ascontiguousarray(g["b_from_y"], np.int32)),
            vm.direct(np.ascontiguousarray(g["block_yx"], np.int32)), g["block_yx"].shape[0], None, None, None)
        if vm.pending() is None:  # (the check may sit in prepare / run rather than in set_lfgroup)
            vm.fn("run", None)(vm.env, self_)
        cls, _ = vm.take()
        assert cls in ("com/traneptora/jxlatte/io/InvalidBitstreamException", "java/lang/IllegalStateException"), cls

        # beginFrame with a buffer shorter than jxl_vardct_params
        vm.fn("beginFrame", None, vp)(vm.env, self_, vm.direct(pbuf, 16))
        assert vm.take()[0] == "java/lang/IllegalArgumentException"
    finally:
        vm.fn("destroy", None, i64)(vm.env, None, handle)

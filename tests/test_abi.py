"""CPU-only: the C-ABI library loads and exports every symbol include/jxlatte_amd.h declares; the
Python mirror of the transform-type table matches the C table; host-side squeeze bookkeeping (no
device needed) matches the oracle and the Python restatement in synth."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from jxlatte_amd import _lib, abi, host, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    header = open(os.path.join(ROOT, "include", "jxlatte_amd.h")).read()
    declared = set(re.findall(r"\b(jxl_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations parsed"
    missing = [n for n in sorted(declared) if not hasattr(lib, n)]
    assert not missing, "symbols declared in the header but not exported: %s" % missing
    # the binding table covers exactly the declared functions
    assert declared == set(_lib.SIGNATURES.keys())
    assert b"gfx950" in lib.jxl_version()


def test_jni_shim_binds_every_declared_symbol():
    """integration/jni cannot be compiled here (no JDK): what CAN be checked is that the C glue calls every function the header
    declares and that the Java class and the C file name the same native methods"""
    header = open(os.path.join(ROOT, "include", "jxlatte_amd.h")).read()
    declared = set(re.findall(r"\b(jxl_[a-z0-9_]+)\s*\(", header))
    c = open(os.path.join(ROOT, "integration", "jni", "jxlatte_amd_jni.c")).read()
    used = set(re.findall(r"\b(jxl_[a-z0-9_]+)\s*\(", c))
    assert not (declared - used), "header entries the JNI glue never calls: %s" % sorted(declared - used)
    java = open(os.path.join(ROOT, "integration", "jni", "NativeBackend.java")).read()
    natives = set(re.findall(r"native\s+[\w\[\]\.]+\s+(\w+)\s*\(", java))
    cfun = set(re.findall(r"Java_com_traneptora_jxlatte_gpu_NativeBackend_(\w+)\(", c))
    assert natives == cfun, (sorted(natives - cfun), sorted(cfun - natives))


def test_frontend_library_exports_every_declared_symbol():
    from jxlatte_amd import frontend
    lib = frontend.load()
    header = open(os.path.join(ROOT, "include", "jxlatte_frontend.h")).read()
    declared = set(re.findall(r"\b(jxf_[a-z0-9_]+)\s*\(", header)) - {"jxf_hooks"}
    assert declared == set(frontend.SIGNATURES.keys())
    assert not [n for n in declared if not hasattr(lib, n)]
    # ctypes mirrors have the C layout (all 4-byte members except the flags word and pointers)
    assert C.sizeof(frontend.ImageInfo) == 4 * (9 + 6 + 2 + 6 + 4 + 9 + 3 + 3 + 1 + 2 + 3 + 5 * 16)
    assert C.sizeof(frontend.Chan) == 24 and C.sizeof(frontend.SqueezeStep) == 16


def test_no_device_fails_loudly():
    """no GPU in the build container: creating a context must fail with a device error, never fall back"""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.JxlError) as e:
        _lib.Context(0)
    assert e.value.status == abi.JXL_ERR_DEVICE


def test_transform_type_table_matches_header():
    hdr = open(os.path.join(ROOT, "include", "jxl_transform_types.h")).read()
    rows = re.findall(r"\{(\d+), (\d+), (\d+), JXL_METHOD_(\w+), (\d+), (\d+)\},\s*/\* (\w+) \*/", hdr)
    assert len(rows) == 27
    methods = {"DCT": 0, "DCT2": 1, "DCT4": 2, "HORNUSS": 3, "DCT8_4": 4, "DCT4_8": 5, "AFV": 6}
    for (t, p, o, m, ph, pw, name), py in zip(rows, abi.TRANSFORM_TYPES):
        assert (name, int(t), int(p), int(o), methods[m], int(ph), int(pw)) == py


def test_struct_sizes_match_c_layout():
    # computed by hand from the header: all members are 4-byte scalars / arrays
    assert C.sizeof(abi.VarDCTParams) == 4 * (3 + 3 + 3 + 1 + 2 + 1 + 1 + 3 + 3 + 1 + 1 + 8 + 3 + 3 + 1 + 9 + 3 + 3 + 1 + 2 + 6)
    assert C.sizeof(abi.SqueezeParam) == 16
    assert C.sizeof(abi.Channel) == 16
    assert C.sizeof(abi.LFGroupDesc) == 16 + 6 * 8 + 8 + 3 * 8


@pytest.mark.parametrize("shape,channels", [((1080, 1920), 3), ((4320, 7680), 3), ((9, 9), 1), ((600, 37), 4), ((8, 8), 3), ((1, 5000), 2)])
def test_default_squeeze_params_three_ways(orc, shape, channels):
    shapes = [shape] * channels
    a = host.ModularStream.defaultSqueezeParams(shapes)       # C-ABI
    b = synth.default_squeeze_params(shapes)                   # Python host logic
    c = orc.default_squeeze_params(shapes)                     # oracle
    assert a == b == c
    ea = host.ModularStream.squeezedShapes(shapes, a)
    assert ea == synth.squeezed_shapes(shapes, a) == orc.squeezed_shapes(shapes, a)
    # inverse bookkeeping returns to the image channels
    assert orc.inverse_shapes(ea, a) == shapes


def test_default_squeeze_plan_1080p_has_18_steps():
    sp = synth.default_squeeze_params([(1080, 1920)] * 3)
    assert len(sp) == 18 and sp[0] == (1, 0, 1, 2) and sp[1] == (0, 0, 1, 2)
    assert len(synth.default_squeeze_params([(4320, 7680)] * 3)) == 22


def test_weights_layout():
    from jxlatte_amd import hfglobal
    w, offs = hfglobal.default_weights()
    assert w.dtype == np.float32 and w.size == 3 * 131584 and offs.shape == (51,)
    assert np.all(np.isfinite(w)) and np.all(w > 0)
    # DCT8 luma DC weight = 1 / 560
    assert w[offs[1]] == np.float32(1.0) / np.float32(560.0)


def test_java_bridge_packs_every_params_field():
    """integration/jni/GpuFrameBridge.packParams writes jxl_vardct_params field by field (all members 4 bytes wide): the number
    of values it puts equals the struct's size in words (no JDK here: the Java source is checked as text)"""
    import ctypes
    import re
    from jxlatte_amd import abi
    src = open(os.path.join(ROOT, "integration", "jni", "GpuFrameBridge.java")).read()
    body = src[src.index("private static ByteBuffer packParams"):src.index("p.flip();")]
    # loops: "for (int X = 0; X < N; X++)" applies to the single statement on the next line
    n = 0
    mult = 1
    for line in body.splitlines():
        code = line.split("//")[0]
        m = re.search(r"for \(int \w+ = 0; \w+ < ([0-9 +]+); \w+\+\+\)", code)
        puts = len(re.findall(r"\.put(?:Int|Float)\(", code))
        if m and not puts:
            mult = sum(int(t) for t in m.group(1).split("+"))
            continue
        n += puts * mult
        if puts:
            mult = 1
    assert n * 4 == ctypes.sizeof(abi.VarDCTParams), (n, ctypes.sizeof(abi.VarDCTParams))


def test_gpu_patch_script_anchors_match_the_reference(tmp_path):
    """tools/patch_reference_for_gpu.sh --patch-only: the call-site patch, the two Frame flags, the three guards (Gaborish, EPF,
    invertXYB) and the three opsin fields land in the reference's sources, each exactly once (skipped where the reference checkout
    is absent, e.g. on the GPU box); with the pin hooks applied first (PIN=1 order) as well. The Java build itself needs a JDK
    and has never run here"""
    import shutil
    import subprocess
    ref = "/root/reference/java"
    if not os.path.isdir(ref) or not shutil.which("perl"):
        pytest.skip("reference checkout or perl absent")
    for pin_first in (False, True):
        j = tmp_path / ("java%d" % pin_first)
        shutil.copytree(ref, str(j))
        if pin_first:
            r = subprocess.run(["bash", os.path.join(ROOT, "tools", "pin_patch_reference.sh"), str(j)], capture_output=True, text=True)
            assert r.returncode == 0, r.stdout + r.stderr
        r = subprocess.run(["bash", os.path.join(ROOT, "tools", "patch_reference_for_gpu.sh"), "--patch-only", str(j)], capture_output=True, text=True)
        assert r.returncode == 0, r.stdout + r.stderr
        base = j / "com" / "traneptora" / "jxlatte"
        out = (base / "frame" / "Frame.java").read_text()
        assert out.count("if (!gpuFrame) passGroup.invertVarDCT(buffers, prev);") == 1
        assert out.count("GpuFrameBridge.invertVarDCT(this, buffers, passGroups, lfGroups, numPasses, numGroups);") == 1
        assert out.index("GpuFrameBridge.enabled(this)") < out.index("if (!gpuFrame)")
        assert out.count("public boolean gpuRestored = false, gpuXYB = false;") == 1
        assert out.count("if (header.restorationFilter.gab && !gpuRestored)\n            performGabConvolution();") == 1
        assert out.count("if (header.restorationFilter.epfIterations > 0 && !gpuRestored)\n            performEdgePreservingFilter();") == 1
        dec = (base / "JXLCodestreamDecoder.java").read_text()
        assert dec.count("if (matrix != null && !frame.gpuXYB)\n            matrix.invertXYB(") == 1
        ops = (base / "color" / "OpsinInverseMatrix.java").read_text()
        for f in ("float[][] matrix;", "float[] opsinBias;", "float[] cbrtOpsinBias;"):
            assert ops.count("    public final " + f) == 1 and ops.count("private final " + f) == 0
        assert (base / "gpu" / "GpuFrameBridge.java").exists() and (base / "gpu" / "NativeBackend.java").exists()
        # what the bridge names in the reference exists there (no JDK: checked as text)
        bridge = (base / "gpu" / "GpuFrameBridge.java").read_text()
        for name, where in (("getGroupLocation", out), ("getPaddedFrameSize", out), ("getLFGroupLocation", out), ("getHFGlobal", out),
                            ("isXYBEncoded", (base / "bundle" / "ImageHeader.java").read_text()),
                            ("getToneMapping", (base / "bundle" / "ImageHeader.java").read_text()),
                            ("saveBeforeCT", (base / "frame" / "FrameHeader.java").read_text())):
            decl = [l for l in where.splitlines() if name in l and "public" in l]
            assert name in bridge and decl, name

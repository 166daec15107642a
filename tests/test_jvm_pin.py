"""Pin-on-arrival: the oracle against planes dumped by the REAL reference (tools/pin_oracle_with_jvm.sh, needs a JDK and a
checkout of jxlatte; neither exists in the build image). While tests/golden/jvm/ holds no dumps the comparison skips -- and the
oracle stays "parity unpinned" (DESIGN.md section 3). With dumps present, every stage of every frame of every sample must match
the reference BIT FOR BIT:
  Modular part of any frame   the stream's channels after applyTransforms (inverse squeeze, RCT: "mod"), the frame buffers after
                              the modular -> buffer conversion ("idct"), after Gaborish / the EPF where a Modular frame has them;
  VarDCT frames               after the inverse transforms ("idct"), invertSubsampling ("sub"), Gaborish ("gab"), the EPF ("epf");
  every frame                 after the colour transform ("xyb");
  the PNG writer              after JXLImage.transform ("tf": transfer function, float) and the planes the IDAT writer reads ("int").
A sample whose dumps exist but which the decoder does not reach (a frame missing on either side) fails, whatever its encoding."""
import glob
import os
import re
import shutil
import struct
import subprocess

import numpy as np
import pytest

from jxlatte_amd import abi

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
JVM = os.path.join(HERE, "golden", "jvm")
SAMPLES = os.path.join(HERE, "golden", "samples")
FRAME_STAGES = ("mod", "idct", "sub", "gab", "epf", "xyb")


def read_dump(path):
    raw = open(path, "rb").read()
    magic, typ, h, w = struct.unpack("<iiii", raw[:16])
    assert magic == 0x3144584A, path
    return np.frombuffer(raw, np.int32 if typ == 0 else np.float32, h * w, 16).reshape(h, w)


def write_dump(path, a):
    a = np.ascontiguousarray(a)
    assert a.dtype in (np.int32, np.float32) and a.ndim == 2, (a.dtype, a.shape)
    with open(path, "wb") as f:
        f.write(struct.pack("<iiii", 0x3144584A, 0 if a.dtype == np.int32 else 1, a.shape[0], a.shape[1]) + a.tobytes())


def dumped_samples():
    """every sample that has ANY dump (frame or PNG stage)"""
    names = set()
    for p in glob.glob(os.path.join(JVM, "*.bin")):
        m = re.match(r"(.+?)\.(f\d+|png)\.[a-z]+\.c\d+\.bin$", os.path.basename(p))
        if m:
            names.add(m.group(1))
    return sorted(names)


def oracle_stage_planes(name):
    """decode `name` with the oracle backend; returns ({(frame, stage): [planes]}, {png stage: [planes]})"""
    from jxlatte_amd.decoder import JXLDecoder, PNGWriter
    from oracle.pybackend import OracleBackend

    frames = {}
    vardct_stages = []  # per VarDCT call: dict stage -> colour planes (stage-masked oracle runs)

    class Capture(OracleBackend):
        def vardct(self, params, weights, woffs, lfgroups, groups):
            groups = list(groups)
            want = params.stages
            out = {}
            for stage, mask in (("idct", abi.STAGE_IDCT), ("gab", abi.STAGE_IDCT | abi.STAGE_GAB),
                                ("epf", abi.STAGE_IDCT | abi.STAGE_GAB | abi.STAGE_EPF)):
                params.stages = mask & want if stage != "idct" else abi.STAGE_IDCT
                out[stage] = np.array(OracleBackend.vardct(self, params, weights, woffs, lfgroups, groups), copy=True)
            # (the oracle's frame entry upsamples subsampled chroma inside the IDCT stage: "idct" of a subsampled frame holds
            # the upsampled planes, which is what the reference has after invertSubsampling -- compared as "sub" only)
            out["sub"] = out["idct"]
            out["subsampled"] = any(params.jpeg_upsampling_y) or any(params.jpeg_upsampling_x)
            params.stages = want
            vardct_stages.append(out)
            return OracleBackend.vardct(self, params, weights, woffs, lfgroups, groups)

    dec = JXLDecoder(os.path.join(SAMPLES, name + ".jxl"), backend=Capture())

    def trace(fi, stage, planes, fused):
        planes = [np.array(p, copy=True) for p in planes]
        if fused and stage in ("idct", "sub", "gab", "epf"):
            # colour planes of a VarDCT frame: from the stage-masked runs of this frame's vardct call (the last one made)
            vs = vardct_stages[-1]
            if stage == "idct" and vs["subsampled"]:
                planes[:3] = [None, None, None]  # not comparable (see Capture.vardct)
            else:
                planes[:3] = [np.ascontiguousarray(vs[stage][c]) for c in range(3)]
        frames[(fi, stage)] = planes

    dec.trace = trace
    image = dec.decode()
    png = {}
    if image is not None:
        hdr = image.isHDR()
        w = PNGWriter(image, bitDepth=16 if hdr else -1, hdr=hdr)
        if not image.has_icc:
            from jxlatte_amd import decoder as D
            t = image.transform(D.PRI_BT2100 if hdr else D.PRI_SRGB, D.WP_D65, D.TF_PQ if hdr else D.TF_SRGB, D.PEAK_DETECT_AUTO)
            png["tf"] = [np.ascontiguousarray(b) for b in t.getBuffer(False)]
        # the writer's samples: rows of interleaved big-endian u8 / u16, colour channels then alpha
        nch = w.colorChannels + (1 if w.alphaIndex >= 0 else 0)
        dt = np.dtype(">u2") if w.bitDepth == 16 else np.dtype("u1")
        s = np.frombuffer(np.ascontiguousarray(w.samples).tobytes(), dt).reshape(w.height, w.width, nch).astype(np.int32)
        chans = {c: s[:, :, c] for c in range(w.colorChannels)}
        if w.alphaIndex >= 0:
            chans[w.colorChannels + w.alphaIndex] = s[:, :, nch - 1]
        png["int"] = chans
    return frames, png


def compare(path, got):
    ref = read_dump(path)
    got = np.asarray(got)
    assert got.shape[0] >= ref.shape[0] and got.shape[1] >= ref.shape[1], (path, got.shape, ref.shape)
    got = got[:ref.shape[0], :ref.shape[1]]
    assert got.dtype == ref.dtype, (path, got.dtype, ref.dtype)
    same = got.view(np.uint32) == ref.view(np.uint32)
    both_nan = (np.isnan(got) & np.isnan(ref)) if got.dtype == np.float32 else np.zeros_like(same)
    assert (same | both_nan).all(), "%s: %d of %d samples differ from the reference" % (path, int((~(same | both_nan)).sum()), ref.size)


def check_sample(name, jvm_dir):
    frames, png = oracle_stage_planes(name)
    checked, unmatched = 0, []
    for path in sorted(glob.glob(os.path.join(jvm_dir, name + ".*.bin"))):
        m = re.match(re.escape(name) + r"\.(f(\d+)|png)\.([a-z]+)\.c(\d+)\.bin$", os.path.basename(path))
        if not m:
            continue
        stage, c = m.group(3), int(m.group(4))
        if m.group(1) == "png":
            planes = png.get(stage)
            got = None if planes is None else (planes.get(c) if isinstance(planes, dict) else (planes[c] if c < len(planes) else None))
            if stage == "int" and got is None and isinstance(planes, dict):
                continue  # an extra channel the PNG writer does not emit (the reference casts every channel, writes colour + alpha)
        else:
            planes = frames.get((int(m.group(2)), stage))
            got = None if planes is None or c >= len(planes) else planes[c]
            if planes is not None and c < len(planes) and got is None:
                continue  # declared not comparable (VarDCT "idct" of a chroma-subsampled frame)
        if got is None:
            unmatched.append(os.path.basename(path))
            continue
        compare(path, got)
        checked += 1
    assert not unmatched, "dumps without a counterpart in the oracle-backed decode: %s" % unmatched[:8]
    assert checked > 0, "no dump of %s was compared" % name
    return checked


@pytest.mark.skipif(not dumped_samples(), reason="no reference dumps under tests/golden/jvm (run tools/pin_oracle_with_jvm.sh on a box with a JDK)")
@pytest.mark.parametrize("name", dumped_samples() or ["none"])
def test_oracle_equals_reference_stage_dumps(name):
    check_sample(name, JVM)


def test_dump_reader_roundtrip(tmp_path):
    """the reader agrees with the writer's format (integration/jvm_pin/StageDump.java): header + row-major 4-byte samples"""
    a = (np.arange(12, dtype=np.float32).reshape(3, 4) - 5) / 3
    p = tmp_path / "x.f0.idct.c0.bin"
    p.write_bytes(struct.pack("<iiii", 0x3144584A, 1, 3, 4) + a.tobytes())
    assert np.array_equal(read_dump(str(p)), a)
    b = np.arange(6, dtype=np.int32).reshape(2, 3) - 2
    p.write_bytes(struct.pack("<iiii", 0x3144584A, 0, 2, 3) + b.tobytes())
    assert np.array_equal(read_dump(str(p)), b)


@pytest.mark.parametrize("name", ["white", "art", "patches-lossless"])
def test_harness_plumbing_on_oracle_made_dumps(tmp_path, name):
    """the comparison machinery end to end, with dumps written from the oracle itself in the reference's file format and under the
    reference's file names (this pins nothing -- it only makes sure that the day real dumps arrive, names, frame numbering, stages,
    shapes and dtypes line up): a VarDCT sample, a Modular sample (squeeze / RCT path, integer planes) and one with several
    frames; frame stages and both PNG stages"""
    frames, png = oracle_stage_planes(name)
    assert frames and png.get("int")
    n = 0
    for (fi, stage), planes in frames.items():
        assert stage in FRAME_STAGES
        for c, a in enumerate(planes):
            if a is None or a.size == 0:
                continue
            write_dump(str(tmp_path / ("%s.f%d.%s.c%d.bin" % (name, fi, stage, c))), a)
            n += 1
    for stage, planes in png.items():
        items = planes.items() if isinstance(planes, dict) else enumerate(planes)
        for c, a in items:
            write_dump(str(tmp_path / ("%s.png.%s.c%d.bin" % (name, stage, c))), np.ascontiguousarray(a))
            n += 1
    assert check_sample(name, str(tmp_path)) == n
    # a dump the decode has no counterpart for (a frame that does not exist) must fail, not pass silently
    write_dump(str(tmp_path / ("%s.f99.idct.c0.bin" % name)), np.zeros((2, 2), np.float32))
    with pytest.raises(AssertionError):
        check_sample(name, str(tmp_path))


def test_pin_patch_anchors_match_the_reference(tmp_path):
    """tools/pin_patch_reference.sh: its eight one-line hooks land in the reference's Frame.java, JXLCodestreamDecoder.java and
    PNGWriter.java, each exactly once and at the statement it is anchored on (skipped where the reference checkout is absent,
    e.g. on the GPU box); the Java build itself needs a JDK and has never run here"""
    ref = "/root/reference/java"
    if not os.path.isdir(ref):
        pytest.skip("reference checkout absent")
    j = tmp_path / "java"
    shutil.copytree(ref, str(j))
    r = subprocess.run(["bash", os.path.join(ROOT, "tools", "pin_patch_reference.sh"), str(j)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    base = j / "com" / "traneptora" / "jxlatte"
    frame = (base / "frame" / "Frame.java").read_text()
    order = [frame.index(s) for s in ('lfGlobal.globalModular.applyTransforms();', 'StageDump.dumpInt("mod", modularBuffer);',
                                      'StageDump.dump("idct", buffer);', 'invertSubsampling();\n', 'StageDump.dump("sub", buffer);',
                                      'performGabConvolution();\n', 'StageDump.dump("gab", buffer);', 'performEdgePreservingFilter();\n',
                                      'StageDump.dump("epf", buffer);')]
    assert order == sorted(order)
    dec = (base / "JXLCodestreamDecoder.java").read_text()
    assert dec.index("performColorTransforms(matrix, frame);") < dec.index('StageDump.dump("xyb", frame.getBuffer());')
    png = (base / "io" / "PNGWriter.java").read_text()
    a, b, c = png.index("image.transform(primaries, whitePoint, tf, peakDetect);"), png.index('StageDump.dumpImage("tf"'), png.index('StageDump.dumpImage("int", buffer);')
    assert a < b < png.index("buffer[c].castToIntWithMax(maxValue);") < c < png.index("public void setWriteSrgbIcc")
    assert (base / "util" / "StageDump.java").exists()

"""Build libjxlatte_amd.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m jxlatte_amd.build [--force]

Flags that matter for parity: -ffp-contract=off (no FMA contraction; the reference rounds every
multiply and add separately), no fast-math, default correctly-rounded f32 division and preserved
f32 denormals.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
SO = os.path.join(HERE, "libjxlatte_amd.so")
SOURCES = ["k_idct.hip", "k_idct_wg3.hip", "k_restore.hip", "k_restore_fused.hip", "k_restore_fused_gen.hip", "k_restore_fused_q.hip", "k_modular.hip", "k_modular_vh.hip", "k_lf.hip", "k_post.hip", "host.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-fno-fast-math",
         "-Wall", "-Wno-unused-function"]
# per-file additions (environment override for experiments: JXL_EXTRA_<stem>="-flag -flag")
# k_restore_fused: hipcc's SLP vectoriser packs the EPF distance terms into v_pk_* pairs but pays for it with ~50 register
# moves and 35 v_and (abs) per channel iteration; scalar code with free |x| source modifiers is 12 % faster (measured).
# k_idct: the IDCT cores use explicit packed vectors; auto-SLP on the 8x8 special transforms costs 19 % (same symptom).
EXTRA = {"k_restore_fused": ["-fno-slp-vectorize"], "k_restore_fused_gen": ["-fno-slp-vectorize"], "k_restore_fused_q": ["-fno-slp-vectorize"], "k_idct": ["-fno-slp-vectorize"], "k_idct_wg3": ["-fno-slp-vectorize"]}


def _deps():
    inc = os.path.join(HERE, "..", "include")
    # sources only: the *.o outputs live in the same directory and must not count as inputs (they did: every call
    # recompiled all but the newest object)
    d = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hip", ".h", ".inc"))]
    d += [os.path.join(inc, f) for f in os.listdir(inc) if f.endswith(".h")]
    return d


def build(force=False, verbose=False, tag=None, defines=()):
    """tag / defines: an experiment build next to the product one -- libjxlatte_amd_<tag>.so with -D<define>... on every file
    (objects under csrc/obj_<tag>/); select it at run time with JXL_AMD_LIB=<path>"""
    newest = max(os.path.getmtime(p) for p in _deps())
    so = SO if not tag else os.path.join(HERE, "libjxlatte_amd_%s.so" % tag)
    objdir = CSRC if not tag else os.path.join(CSRC, "obj_" + tag)
    os.makedirs(objdir, exist_ok=True)
    objs = []
    jobs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        objs.append(obj)
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < newest:
            stem = src.replace(".hip", "")
            extra = os.environ.get("JXL_EXTRA_" + stem, None)
            extra = extra.split() if extra is not None else EXTRA.get(stem, [])
            jobs.append([HIPCC] + FLAGS + extra + ["-D" + d for d in defines] + ["-c", os.path.join(CSRC, src), "-o", obj])
    if jobs:
        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stderr[-6000:]))
            return r.stderr
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for err in ex.map(run, jobs):
                if verbose and err.strip():
                    print(err)
    if jobs or not os.path.exists(so):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    return so


def build_stream_bench(force=False):
    """the native host-thread driver of bench.py's streaming boundary leg (tools/native/stream_bench.cpp; g++, no device code):
    a measurement harness beside the library, reaching it through dlopen"""
    src = os.path.join(HERE, "..", "tools", "native", "stream_bench.cpp")
    so = os.path.join(HERE, "libjxl_stream_bench.so")
    inc = os.path.join(HERE, "..", "include")
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(src), os.path.getmtime(os.path.join(inc, "jxlatte_amd.h"))):
        cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-pthread", "-Wall", "-I", inc, src, "-o", so, "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("stream_bench build failed:\n%s" % r.stderr[-4000:])
    return so


if __name__ == "__main__":
    tag = next((a.split("=", 1)[1] for a in sys.argv if a.startswith("--tag=")), None)
    print(build(force="--force" in sys.argv, verbose=True, tag=tag, defines=[a[2:] for a in sys.argv if a.startswith("-D")]))
    if not tag:
        print(build_stream_bench(force="--force" in sys.argv))

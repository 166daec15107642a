"""what would FMA contraction in the restoration kernels buy, and what would it cost in ulp?

    JXL_EXTRA_k_restore_fused="-fno-slp-vectorize -ffp-contract=fast" JXL_EXTRA_k_restore="-ffp-contract=fast" \
        python -m jxlatte_amd.build --tag=fastfp                      # experiment library (IDCT files keep contract=off)
    JXL_AMD_LIB=jxlatte_amd/libjxlatte_amd_fastfp.so python tools/fast_fp.py > gpurun_out/fast_fp.json   (GPU box)
    python tools/fast_fp.py                                              # the product library: every histogram must be {0: all}

For every parity frame (synthetic mixes, all stage masks that include Gab / EPF / XYB) and the real VarDCT samples: the
distribution of the ulp distance between the library's float result and the oracle's, and the frame time."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from jxlatte_amd import _lib, abi, host, synth
from oracle import pyoracle as orc


def ordered(a):
    i = np.ascontiguousarray(a, np.float32).view(np.int32).astype(np.int64)
    return np.where(i < 0, -(i & 0x7fffffff), i)


BINS = [0, 1, 2, 4, 16, 256, 65536, 1 << 62]


def hist(got, exp):
    d = np.abs(ordered(got) - ordered(exp))
    nan = np.isnan(got) | np.isnan(exp)
    d = d[~nan]
    out = {"n": int(d.size), "max_ulp": int(d.max()), "identical": float((d == 0).mean()), "le1": float((d <= 1).mean()),
           "max_abs": float(np.abs(got.astype(np.float64) - exp.astype(np.float64))[~nan].max()),
           "nan_mismatch": int((np.isnan(got) != np.isnan(exp)).sum())}
    h = {}
    for lo, hi in zip(BINS[:-1], BINS[1:]):
        h["%d..%d" % (lo, hi - 1) if hi - lo > 1 else str(lo)] = int(((d >= lo) & (d < hi)).sum())
    out["hist"] = h
    return out


def main():
    orc.lib()
    ctx = _lib.Context(0)
    res = {"lib": _lib.SO_PATH, "frames": {}}
    S = abi
    masks = {"idct+gab": S.STAGE_IDCT | S.STAGE_GAB, "idct+epf": S.STAGE_IDCT | S.STAGE_EPF,
             "idct+gab+epf": S.STAGE_IDCT | S.STAGE_GAB | S.STAGE_EPF, "idct+xyb": S.STAGE_IDCT | S.STAGE_XYB,
             "all": S.STAGE_IDCT | S.STAGE_GAB | S.STAGE_EPF | S.STAGE_XYB}
    for (w, h, mix, seed) in [(512, 512, "default", 1), (520, 264, "default", 2), (1024, 1024, "large", 3), (3840, 2160, "default", 1000)]:
        fr = synth.make_vardct_frame(w, h, seed=seed, mix=mix, aligned=False)
        for name, st in masks.items():
            if w > 2000 and name != "all":
                continue
            f = host.Frame.from_synth(ctx, fr, stages=st)
            got = f.decodeFrame()
            t0 = time.perf_counter()
            for _ in range(20):
                f.run()
            ctx.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            r = hist(got, orc.vardct_frame(fr, stages=st))
            r["ms_per_frame"] = round(ms, 4)
            res["frames"]["%dx%d %s seed %d: %s" % (w, h, mix, seed, name)] = r
    # real bitstreams, whole decode (device backend vs oracle backend)
    from jxlatte_amd.decoder import DeviceBackend, JXLDecoder
    from oracle.pybackend import OracleBackend
    dev, ob = DeviceBackend(0), OracleBackend()
    sdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "samples")
    for name in ("bbb", "lenna", "white"):
        p = os.path.join(sdir, name + ".jxl")
        got = JXLDecoder(p, backend=dev).decode()
        exp = JXLDecoder(p, backend=ob).decode()
        res["frames"][name + ".jxl"] = hist(np.stack(got.buffer[:3]), np.stack(exp.buffer[:3]))
    json.dump(res, sys.stdout, indent=1)
    print()


if __name__ == "__main__":
    main()

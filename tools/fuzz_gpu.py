"""Randomised device-vs-oracle parity sweep (GPU box): frame sizes off every tile grid, varblock mixes, aligned / unaligned
tilings, Gaborish on / off, 0-3 EPF iterations, float / u8 / u16 / interleaved outputs, sRGB / PQ, single runs and batches.
    python tools/fuzz_gpu.py [n_cases] [seed]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jxlatte_amd import _lib, abi, host, synth
from oracle import pyoracle as orc


def run(n_cases=40, seed=2026, device=0, verbose=True):
    """returns the number of mismatching frames"""
    rng = np.random.default_rng(seed)
    ctxs = [_lib.Context(device) for _ in range(4)]
    bad = 0
    try:
        for case in range(n_cases):
            batch = int(rng.integers(1, 5))
            frames, synths, exps = [], [], []
            for k in range(batch):
                w, h = int(rng.integers(1, 60)) * 8, int(rng.integers(1, 40)) * 8
                mix = ["default", "all", "dct8", "large"][int(rng.integers(0, 4))]
                if mix == "large":
                    w, h = max(w, 256), max(h, 256)
                kw = dict(epf_iters=int(rng.integers(0, 4)), nonzero_p=float(rng.choice([0.0, 0.02, 0.15, 0.5])))
                fmt = int(rng.integers(0, 5))
                if fmt == 1:
                    kw.update(transfer=abi.TRANSFER_SRGB, out_format=abi.OUT_U8)
                elif fmt == 2:
                    kw.update(transfer=abi.TRANSFER_PQ, out_format=abi.OUT_U16)
                elif fmt == 3:
                    kw.update(transfer=abi.TRANSFER_SRGB, out_format=abi.OUT_RGB8)
                fr = synth.make_vardct_frame(w, h, seed=int(rng.integers(1, 1 << 30)), mix=mix, aligned=bool(rng.integers(0, 2)), **kw)
                if rng.integers(0, 4) == 0:
                    fr["params"].gab = 0
                synths.append(fr)
                frames.append(host.Frame.from_synth(ctxs[k], fr))
                exps.append(orc.vardct_frame(fr))
            if batch == 1:
                frames[0].run()
            else:
                host.Frame.runBatch(frames)
            for k, fr in enumerate(frames):
                got = fr.readOutput()
                exp = exps[k]
                if got.dtype == np.float32:
                    ok = np.array_equal(got.view(np.uint32), exp.view(np.uint32))
                else:
                    e = exp if exp.shape == got.shape else np.moveaxis(exp, 0, -1)
                    ok = np.abs(got.astype(np.int64) - e.astype(np.int64)).max() <= 1  # transfer stage: <= 1 code value
                if not ok:
                    bad += 1
                    p = synths[k]["params"]
                    print("MISMATCH case %d frame %d: %dx%d mix=%s epf=%d gab=%d fmt=%d batch=%d" % (case, k, p.width, p.height, synths[k]["mix"], p.epf_iters, p.gab, p.out_format, batch))
    finally:
        for c in ctxs:
            c.close()
    if verbose:
        print("fuzz: %d cases, %d mismatching frames" % (n_cases, bad))
    return bad


if __name__ == "__main__":
    sys.exit(1 if run(int(sys.argv[1]) if len(sys.argv) > 1 else 40, int(sys.argv[2]) if len(sys.argv) > 2 else 2026) else 0)

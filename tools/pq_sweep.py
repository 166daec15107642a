"""Exhaustive check of the device's PQ transfer (GPU box): ALL 2^32 float32 bit patterns through jxl_stage_transfer (the
tabulated fast path inside [2^-40, 4), the double-precision form outside) against the oracle's TF_PQ.fromLinear
(TransferFunction.java:83-87: two double pows, cast to float). Reports the ulp-difference histogram; exits non-zero
if any input differs by more than 1 ulp or in NaN-ness.

    python tools/pq_sweep.py [--chunk-log2 26] [--out profiles/r2_pq_sweep.txt]
"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from jxlatte_amd import _lib, abi, host
from oracle import pyoracle as orc

ap = argparse.ArgumentParser()
ap.add_argument("--chunk-log2", type=int, default=26)
ap.add_argument("--out", default="")
ap.add_argument("--limit", type=int, default=0, help="only the first N chunks (smoke)")
ap.add_argument("--srgb8", action="store_true", help="instead: sRGB + 8-bit quantisation (the threshold table fp_srgb8) against the oracle, exact equality")
ap.add_argument("--srgb16", action="store_true", help="instead: sRGB + 16-bit quantisation (fp_srgb16) against the oracle, exact equality")
ap.add_argument("--pq8", action="store_true", help="instead: PQ + 8-bit quantisation (fp_pq8) against the oracle, exact equality")
ap.add_argument("--pq16", action="store_true", help="instead: PQ + 16-bit quantisation (fp_pq16: table + threshold correction) against the oracle, exact equality")
args = ap.parse_args()
if args.srgb8 or args.pq16 or args.srgb16 or args.pq8:
    TF, MAXV, NAME = (abi.TRANSFER_SRGB, 255, "sRGB + castToIntWithMax(255), device (threshold table)") if args.srgb8 else \
                     (abi.TRANSFER_SRGB, 65535, "sRGB + castToIntWithMax(65535), device (table + thresholds)") if args.srgb16 else \
                     (abi.TRANSFER_PQ, 255, "PQ + castToIntWithMax(255), device (thresholds)") if args.pq8 else \
                     (abi.TRANSFER_PQ, 65535, "PQ + castToIntWithMax(65535), device (table + thresholds)")
    ctx = _lib.Context(0)
    n = 1 << args.chunk_log2
    chunks = (1 << 32) // n if not args.limit else min((1 << 32) // n, args.limit)
    bad, first = 0, None
    hist = np.zeros(MAXV + 1, np.int64)
    t0 = time.time()
    for k in range(chunks):
        bits = (np.arange(n, dtype=np.uint64) + np.uint64(k) * np.uint64(n)).astype(np.uint32)
        x = bits.view(np.float32)
        got = host.transfer(ctx, x, TF, MAXV)
        exp = orc.transfer(x, TF, MAXV)
        ne = np.flatnonzero(got != exp)
        bad += ne.size
        if ne.size and first is None:
            first = (int(bits[ne[0]]), int(got[ne[0]]), int(exp[ne[0]]))
        hist += np.bincount(exp.astype(np.int64), minlength=MAXV + 1)[:MAXV + 1]
        if k % 8 == 7:
            print("chunk %d / %d  mismatches %d  (%.0f s)" % (k + 1, chunks, bad, time.time() - t0), flush=True)
    lines = ["%s vs oracle, all float32 inputs in %d chunks of 2^%d (%s)"
             % (NAME, chunks, args.chunk_log2, _lib.load().jxl_version().decode()),
             "inputs compared: %d" % (chunks * n), "mismatches: %d" % bad,
             "first mismatch (input bits, device, oracle): %s" % (first,),
             "inputs per level: 0 -> %d, %d -> %d, between -> %d; levels never produced: %d" % (hist[0], MAXV, hist[MAXV], hist[1:MAXV].sum(), int((hist == 0).sum())),
             "wall time %.0f s" % (time.time() - t0)]
    print("\n".join(lines))
    if args.out:
        os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
        open(args.out, "w").write("\n".join(lines) + "\n")
    sys.exit(0 if bad == 0 else 1)
ctx = _lib.Context(0)
n = 1 << args.chunk_log2
chunks = (1 << 32) // n
if args.limit:
    chunks = min(chunks, args.limit)
hist = np.zeros(4, np.int64)  # 0, 1, 2, >2 ulp
nan_mismatch = 0
worst = (0, 0, 0.0, 0.0)
t0 = time.time()
for k in range(chunks):
    bits = (np.arange(n, dtype=np.uint64) + np.uint64(k) * np.uint64(n)).astype(np.uint32)
    x = bits.view(np.float32)
    got = host.transfer(ctx, x, abi.TRANSFER_PQ)
    exp = orc.transfer(x, abi.TRANSFER_PQ)
    gn, en = np.isnan(got), np.isnan(exp)
    nan_mismatch += int((gn != en).sum())
    ok = ~(gn | en)
    gi = got.view(np.int32).astype(np.int64)
    ei = exp.view(np.int32).astype(np.int64)
    # monotone integer key of a float (negative floats mirrored), so that the difference counts ulps across zero too
    gi = np.where(gi < 0, -(gi & 0x7FFFFFFF), gi)
    ei = np.where(ei < 0, -(ei & 0x7FFFFFFF), ei)
    d = np.abs(gi - ei)[ok]
    hist += np.bincount(np.minimum(d, 3), minlength=4)[:4]
    if d.size and d.max() > worst[0]:
        j = np.flatnonzero(ok)[int(d.argmax())]
        worst = (int(d.max()), int(bits[j]), float(got[j]), float(exp[j]))
    if k % 8 == 7:
        print("chunk %d / %d  hist %s  nan mismatches %d  (%.0f s)" % (k + 1, chunks, hist.tolist(), nan_mismatch, time.time() - t0), flush=True)
total = int(hist.sum())
lines = ["PQ transfer, device vs oracle, all float32 inputs in %d chunks of 2^%d (%s)" % (chunks, args.chunk_log2, _lib.load().jxl_version().decode()),
         "non-NaN outputs compared: %d" % total,
         "identical: %d (%.6f %%)" % (hist[0], 100.0 * hist[0] / max(total, 1)),
         "1 ulp apart: %d (%.6f %%)" % (hist[1], 100.0 * hist[1] / max(total, 1)),
         "2 ulp apart: %d" % hist[2], "more than 2 ulp apart: %d" % hist[3],
         "NaN on one side only: %d" % nan_mismatch,
         "worst: %d ulp at input bits 0x%08x (device %r, oracle %r)" % worst,
         "wall time %.0f s" % (time.time() - t0)]
print("\n".join(lines))
if args.out:
    os.makedirs(os.path.dirname(args.out) or ".", exist_ok=True)
    open(args.out, "w").write("\n".join(lines) + "\n")
sys.exit(0 if hist[2] == 0 and hist[3] == 0 and nan_mismatch == 0 else 1)

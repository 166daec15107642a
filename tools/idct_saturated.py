"""IDCT stage per transform type in the SATURATED regime: N contexts, each with a 4K frame of ONE type (or a named mix), all in flight
on their own streams (what bench.py's timed step does), wall clock over K steps -> ms per frame, beside the type's byte floor
(24 B/px at the streaming rate).

    python tools/idct_saturated.py [-n 8] [--steps 20] [TYPE | MIX ...]      MIX = "DCT8=0.5+DCT16=0.5" or a name of synth.MIXES

VERDICT r5 item 1(a): the table that names the type which drags the default mix from 4.7 to 2.6 TB/s."""
import argparse
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from jxlatte_amd import _lib, abi, host, synth  # noqa: E402

TYPES = ["DCT8", "DCT16", "DCT32", "DCT16_8", "DCT8_16", "DCT32_8", "DCT8_32", "DCT32_16", "DCT16_32", "DCT64", "DCT64_32", "DCT32_64",
         "DCT4", "DCT4_8", "DCT8_4", "DCT2", "HORNUSS", "AFV0", "default"]
W, H = 3840, 2160
HBM_STREAM = 6.3e12   # plain streaming rate measured on this part (tools/ubench/vh_pattern.hip), B/s
VALU_RATE = 1228.8e9  # wave-instructions per second at the 2-cycle issue peak


def run(spec, n, steps, ctxs):
    mix = spec if ("=" in spec or spec in synth.MIXES) else "%s=1.0" % spec
    frames = []
    fr0 = None
    for i in range(n):
        if i < 2:
            fr0 = synth.make_vardct_frame(W, H, seed=1000 + i, mix=mix)
            keep = fr0 if i == 0 else keep
        frames.append(host.Frame.from_synth(ctxs[i], fr0, stages=abi.STAGE_IDCT))
    hist = synth.type_histogram(keep)
    for _ in range(3):
        for f in frames:
            f.run()
    for c in ctxs[:n]:
        c.synchronize()
    reps = []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(steps):
            for f in frames:
                f.run()
        for c in ctxs[:n]:
            c.synchronize()
        reps.append((time.perf_counter() - t0) / (steps * n))
    launches = frames[0].lastLaunchCount()
    return float(np.median(reps)), launches, hist


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("-n", type=int, default=8)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--also-single", action="store_true", help="a second column: one frame alone")
    ap.add_argument("specs", nargs="*")
    a = ap.parse_args()
    ctxs = [_lib.Context(0) for _ in range(a.n)]
    byte_floor = 24.0 * W * H / HBM_STREAM * 1e6
    print("# IDCT stage only (stages=1), 3840x2160, %d frames in flight, median of 7 x %d steps; byte floor %.1f us per frame (24 B/px at %.1f TB/s)"
          % (a.n, a.steps, byte_floor, HBM_STREAM / 1e12))
    print("%-28s %10s %10s %10s %9s" % ("type / mix", "us/frame", "TB/s alg", "x floor", "launches"))
    for spec in a.specs or TYPES:
        s, launches, hist = run(spec, a.n, a.steps, ctxs)
        line = "%-28s %10.1f %10.2f %10.2f %9d" % (spec, s * 1e6, 24.0 * W * H / s / 1e12, s * 1e6 / byte_floor, launches)
        if a.also_single:
            s1, _, _ = run(spec, 1, a.steps, ctxs)
            line += "   alone %.1f us" % (s1 * 1e6)
        print(line, flush=True)


if __name__ == "__main__":
    main()

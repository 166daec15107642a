"""Is bench.py's timed step bound by the host (the time the run() calls take to return) or by the device? Per step of N 4K frames: issue time
(all jxl_vardct_run calls returned) against completion time (contexts synchronised), for the whole path and per stage mask.

    python tools/host_bound_check.py [-n 8] [--steps 20]"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
from jxlatte_amd import _lib, abi, host, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=8)
ap.add_argument("--steps", type=int, default=20)
a = ap.parse_args()
ctxs = [_lib.Context(0) for _ in range(a.n)]
for stages, name in ((31, "whole path"), (1, "IDCT only"), (30, "restoration only")):
    frames = []
    for i in range(a.n):
        fr = synth.make_vardct_frame(3840, 2160, seed=1000 + (i % 2), mix="default")
        frames.append(host.Frame.from_synth(ctxs[i], fr, stages=stages))
    for _ in range(3):
        for f in frames:
            f.run()
    for c in ctxs:
        c.synchronize()
    iss, tot = [], []
    for _ in range(7):
        t0 = time.perf_counter()
        for _ in range(a.steps):
            for f in frames:
                f.run()
        t1 = time.perf_counter()
        for c in ctxs:
            c.synchronize()
        t2 = time.perf_counter()
        iss.append((t1 - t0) / a.steps)
        tot.append((t2 - t0) / a.steps)
    print("%-18s per step of %d frames: issue %.3f ms, complete %.3f ms  (per frame: issue %.1f us, complete %.1f us)" %
          (name, a.n, np.median(iss) * 1e3, np.median(tot) * 1e3, np.median(iss) * 1e6 / a.n, np.median(tot) * 1e6 / a.n))

"""Host time of jxl_vardct_run (enqueue only) against the device time of the same frames: is a single stream of frames host-bound?
   python tools/r3_host_enqueue.py [n]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jxlatte_amd import _lib, host, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
ctx = _lib.Context(0)
fr = host.Frame.from_synth(ctx, synth.make_vardct_frame(3840, 2160, seed=1000, mix="default"))
for _ in range(30):
    fr.run()
ctx.synchronize()
for rep in range(3):
    per = []
    t0 = time.perf_counter()
    for _ in range(n):
        a = time.perf_counter()
        fr.run()
        per.append((time.perf_counter() - a) * 1e6)
    t1 = time.perf_counter()
    ctx.synchronize()
    t2 = time.perf_counter()
    print("rep %d: enqueue of %d frames %.1f us per frame (median call %.1f, min %.1f, max %.1f); until the device is done %.1f us per frame"
          % (rep, n, (t1 - t0) * 1e6 / n, np.median(per), min(per), max(per), (t2 - t0) * 1e6 / n))
lat = []
for _ in range(10):
    ctx.synchronize()
    a = time.perf_counter(); fr.run(); b = time.perf_counter(); ctx.synchronize(); c = time.perf_counter()
    lat.append(((b - a) * 1e6, (c - a) * 1e6))
print("one frame at a time: run() returns after %.1f us, synchronize() after %.1f us (medians)" % (np.median([x[0] for x in lat]), np.median([x[1] for x in lat])))

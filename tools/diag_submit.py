"""is the batch step bound by host-side launch cost?  submit time (no sync) vs total time per step"""
import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from jxlatte_amd import _lib, host, synth
st = int(sys.argv[1]) if len(sys.argv) > 1 else 31
fpg = int(sys.argv[2]) if len(sys.argv) > 2 else 8
fr0 = [synth.make_vardct_frame(3840, 2160, seed=1000 + i, mix="default") for i in range(2)]
ctxs = [_lib.Context(0) for _ in range(fpg)]
frames = [host.Frame.from_synth(c, fr0[i % 2], stages=st) for i, c in enumerate(ctxs)]
def sync():
    for c in ctxs: c.synchronize()
for _ in range(3):
    for f in frames: f.run()
sync()
K = 20
t0 = time.perf_counter()
for _ in range(K):
    for f in frames: f.run()
t1 = time.perf_counter()
sync()
t2 = time.perf_counter()
print("stages=%d fpg=%d launches/frame=%d: submit %.3f ms/step, total %.3f ms/step -> %.1f Mpx/s" % (
    st, fpg, frames[0].lastLaunchCount(), (t1 - t0) * 1e3 / K, (t2 - t0) * 1e3 / K, 3840 * 2160 * fpg * K / (t2 - t0) / 1e6))

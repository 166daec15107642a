"""CPU-side enqueue cost of jxl_vardct_run vs GPU time per 4K frame"""
import sys, time, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
from jxlatte_amd import _lib, abi, host, synth
ctx = _lib.Context(0)
fr = synth.make_vardct_frame(3840, 2160, seed=1000)
f = host.Frame.from_synth(ctx, fr)
for _ in range(3): f.run()
ctx.synchronize()
N = 50
t0 = time.perf_counter()
for _ in range(N): f.run()
t1 = time.perf_counter()
ctx.synchronize()
t2 = time.perf_counter()
print("enqueue %.1f us/frame, total %.1f us/frame, launches %d" % ((t1 - t0) / N * 1e6, (t2 - t0) / N * 1e6, f.lastLaunchCount()))

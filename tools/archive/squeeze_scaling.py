"""duration of one inverse H / V squeeze launch vs the number of independent chains (rows / columns)"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jxlatte_amd import _lib, host
ctx = _lib.Context(0)
rng = np.random.default_rng(1)
for h in (64, 256, 1024, 2048, 4320, 8640):
    avg = rng.integers(0, 255, (h, 3840)).astype(np.int32)
    res = rng.integers(-8, 9, (h, 3840)).astype(np.int32)
    ms = host.ModularStream(ctx, [avg, res], [(1, 1, 0, 1)])
    ms.begin()
    for _ in range(2): ms.run()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): ms.run()
    ctx.synchronize()
    print("H-squeeze 3840 pairs x %5d rows: %.1f us per launch" % (h, (time.perf_counter() - t0) / 5 * 1e6))
for w in (64, 1024, 7680, 15360):
    avg = rng.integers(0, 255, (2160, w)).astype(np.int32)
    res = rng.integers(-8, 9, (2160, w)).astype(np.int32)
    ms = host.ModularStream(ctx, [avg, res], [(0, 1, 0, 1)])
    ms.begin()
    for _ in range(2): ms.run()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): ms.run()
    ctx.synchronize()
    print("V-squeeze 2160 pairs x %5d cols: %.1f us per launch" % (w, (time.perf_counter() - t0) / 5 * 1e6))

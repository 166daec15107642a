"""IDCT-stage time (HIP events) for single-type 4K frames: where does the IDCT stage spend its time? ("DCT8" = a frame of that type only, "DCT8:0.7,DCT32:0.3" = a mix)"""
import sys, ctypes as C
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from jxlatte_amd import _lib, abi, host, synth
ctx = _lib.Context(0)
W, H = 3840, 2160
for name in sys.argv[1:] or ["DCT8", "DCT16", "DCT32", "DCT64", "DCT16_8", "DCT8_32", "DCT32_16", "DCT64_32", "AFV0", "DCT4", "HORNUSS"]:
    # "DCT8" = a frame of that type only; "DCT8:0.7,DCT32:0.3" = a custom mix (area shares)
    mix = {k: float(v) for k, v in (kv.split(":") for kv in name.split(","))} if ":" in name else {name: 1.0}
    fr = synth.make_vardct_frame(W, H, seed=1, mix=mix, nonzero_p=0.15)
    f = host.Frame.from_synth(ctx, fr, stages=abi.STAGE_IDCT)
    for _ in range(3): f.run()
    ctx.synchronize()
    ctx.call("jxl_vardct_enable_stage_timing", 1)
    for _ in range(10): f.run()
    ms = C.c_float()
    ctx.call("jxl_vardct_last_stage_ms", 1, C.byref(ms))
    ctx.call("jxl_vardct_enable_stage_timing", 0)
    hist = synth.type_histogram(fr)
    print("%-10s idct_stage %.1f us  (share of that type %.2f, launches %d)" % (name, ms.value * 1e3, hist.get(name, 0) if ":" not in name else sum(hist.get(k, 0) for k in mix), f.lastLaunchCount()))

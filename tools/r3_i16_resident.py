"""VERDICT r2 item 7 as an experiment: the ≤ 64-point IDCT launches reading the committed int16 coefficient planes directly
(JXL_WG3_I16=1) against the int32 planes k_widen2d makes of them. One 4K frame through map_coeffs_i16 / commit, then the frame
alone on the device: stage time by the library's events, output compared with the int32 run of the same process order.
   python tools/r3_i16_resident.py            (run once with JXL_WG3_I16=1 and once without; the env is read once per process)"""
import ctypes as C, os, sys, time, hashlib
os.environ.setdefault("JXL_COMMIT_ZEROCOPY", "0")  # the experiment reads the staged int16 planes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from jxlatte_amd import _lib, host, synth
ctx = _lib.Context(0)
d = synth.make_vardct_frame(3840, 2160, seed=1000, mix=sys.argv[1] if len(sys.argv) > 1 else "default")
fr = host.Frame(ctx, d["params"], d["weights"], d["woffs"])
for g in d["lfgroups"]:
    fr.setLFGroup(g)
mp = fr.mapCoeffsI16()
for ch in range(3):
    np.copyto(mp[ch], np.ascontiguousarray(d["coeff"][ch], np.int16))
fr.commitCoeffsI16()
for _ in range(100):
    fr.run()
ctx.call("jxl_vardct_enable_stage_timing", 1)
for _ in range(32):
    fr.run()
ctx.synchronize()
v = C.c_float()
out = {}
for which, nm in ((0, "frame"), (1, "idct"), (2, "restore")):
    ctx.call("jxl_vardct_last_stage_ms", which, C.byref(v))
    out[nm] = v.value * 1e3
ctx.call("jxl_vardct_enable_stage_timing", 0)
lat = []
for _ in range(20):
    ctx.synchronize()
    a = time.perf_counter(); fr.run(); ctx.synchronize(); lat.append((time.perf_counter() - a) * 1e6)
px = fr.readOutput()
h = hashlib.sha256(b"".join(np.ascontiguousarray(p).tobytes() for p in px)).hexdigest()[:16]
print("JXL_WG3_I16=%s: IDCT stage %.1f us, restoration %.1f us, frame %.1f us (events, back-to-back runs); one frame at a time %.1f us; output sha %s"
      % (os.environ.get("JXL_WG3_I16", "0"), out["idct"], out["restore"], out["frame"], float(np.median(lat)), h))

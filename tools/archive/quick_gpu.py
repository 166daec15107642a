import sys, time
sys.path.insert(0, '.')
import numpy as np
from jxlatte_amd import abi, host, synth, _lib
from oracle import pyoracle as orc
ctx = _lib.Context(0)
print(_lib.load().jxl_version())
for mix, al, seed in (("dct8", True, 4), ("all", True, 1), ("all", False, 3), ("large", True, 5)):
    W, H = (1024, 512) if mix == "large" else (512, 256)
    frame = synth.make_vardct_frame(W, H, seed=seed, mix=mix, aligned=al)
    for name, st in (("idct", 1), ("gab", 3), ("epf", 7), ("all", 15)):
        fr = host.Frame.from_synth(ctx, frame, stages=st)
        got = fr.decodeFrame()
        exp = orc.vardct_frame(frame, stages=st)
        bad = (got.view(np.uint32) != exp.view(np.uint32))
        print(mix, al, name, "launches", fr.lastLaunchCount(), "mismatch", int(bad.sum()), "of", bad.size, "maxabs", float(np.nanmax(np.abs(got - exp))))
        if bad.any() and name == "idct":
            ys, xs = np.nonzero(bad.any(axis=0))
            cells = set(zip((ys // 8).tolist(), (xs // 8).tolist()))
            types = {}
            for (cy, cx) in cells:
                t = int(frame["dct_select"][cy, cx]); types[t] = types.get(t, 0) + 1
            print("   bad cells by type:", {abi.TT_NAME[t]: n for t, n in types.items()})

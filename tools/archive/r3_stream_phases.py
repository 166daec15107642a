"""Where a frame's host time goes when N contexts stream frames through the whole boundary from N host threads
(the `untimed.streaming` leg of bench.py, with a timer around every call):  python tools/r3_stream_phases.py [n_ctx ...]"""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("R3_WITH_TORCH"):  # bench.py's process has torch loaded (its streaming leg reads lower: is it torch?)
    import torch
    torch.zeros(1, device="cuda")
from jxlatte_amd import _lib, abi, host, synth

W, H = 3840, 2160
d = synth.make_vardct_frame(W, H, seed=1000, mix="default")
p = d["params"]
p.transfer, p.out_format = abi.TRANSFER_SRGB, abi.OUT_RGB8
lib = _lib.load()
coeff16 = [np.ascontiguousarray(a, np.int16) for a in d["coeff"]]
PH = ["begin+weights", "lfgroups", "prepare", "map", "stores", "commit", "run", "read_output"]


def leg(n_ctx, frames_per_ctx=6):
    ctxs = [_lib.Context(0) for _ in range(n_ctx)]
    pouts = [host.PinnedArray(lib, (H, W * 3), np.uint8) for _ in range(n_ctx)]
    acc = np.zeros((n_ctx, len(PH)))
    start = threading.Barrier(n_ctx + 1)

    def worker(i):
        c = ctxs[i]
        pp = (C.c_void_p * 3)(pouts[i].array.ctypes.data, None, None)

        def one(rec):
            t = [time.perf_counter()]
            fr = host.Frame(c, p, d["weights"], d["woffs"]); t.append(time.perf_counter())
            for g in d["lfgroups"]:
                fr.setLFGroup(g)
            t.append(time.perf_counter())
            c.call("jxl_vardct_prepare"); t.append(time.perf_counter())
            mp = fr.mapCoeffsI16(); t.append(time.perf_counter())
            for ch in range(3):
                np.copyto(mp[ch], coeff16[ch])
            t.append(time.perf_counter())
            fr.commitCoeffsI16(); t.append(time.perf_counter())
            fr.run(); t.append(time.perf_counter())
            c.call("jxl_vardct_read_output", pp, fr.width); t.append(time.perf_counter())
            if rec:
                acc[i] += np.diff(t)
        one(False)
        start.wait()
        for _ in range(frames_per_ctx):
            one(True)

    th = [threading.Thread(target=worker, args=(i,)) for i in range(n_ctx)]
    for t in th:
        t.start()
    start.wait()
    a = time.perf_counter()
    for t in th:
        t.join()
    wall = time.perf_counter() - a
    for x in pouts:
        x.free()
    for c in ctxs:
        c.close()
    n = n_ctx * frames_per_ctx
    per = acc.sum(axis=0) / n * 1e3
    print("%2d contexts: %.2f ms per frame wall (%.0f Mpx/s); per frame and thread, ms: %s  = %.2f"
          % (n_ctx, wall * 1e3 / n, W * H * n / wall / 1e6, ", ".join("%s %.2f" % (k, v) for k, v in zip(PH, per)), per.sum()), flush=True)


for n in [int(a) for a in sys.argv[1:]] or [1, 2, 3, 6]:
    leg(n)

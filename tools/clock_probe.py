"""The shader clock the chip holds under each stage of the path (jxl_debug_clock_probe: one wave counting s_memtime against the 100 MHz
s_memrealtime while N frames run): is the batch bound by the package power limit?

    python tools/clock_probe.py [-n 8]"""
import argparse, ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from jxlatte_amd import _lib, abi, host, synth  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("-n", type=int, default=8)
a = ap.parse_args()
lib = _lib.load()
lib.jxl_debug_clock_probe.restype = C.c_int
lib.jxl_debug_clock_probe.argtypes = [C.c_int, C.c_double, C.POINTER(C.c_double)]


def probe(us=2000.0):
    v = C.c_double()
    assert lib.jxl_debug_clock_probe(0, us, C.byref(v)) == 0
    return v.value


ctxs = [_lib.Context(0) for _ in range(a.n)]
print("idle chip: %.0f MHz" % probe())
for stages, name in ((31, "whole path"), (1, "IDCT only"), (30, "restoration only")):
    frames = []
    for i in range(a.n):
        fr = synth.make_vardct_frame(3840, 2160, seed=1000 + (i % 2), mix="default")
        frames.append(host.Frame.from_synth(ctxs[i], fr, stages=stages))
    stop = False

    def feed():
        while not stop:
            for _ in range(10):
                for f in frames:
                    f.run()
            for c in ctxs:
                c.synchronize()

    th = threading.Thread(target=feed)
    th.start()
    time.sleep(1.0)  # a second of load first: the clock settles
    vals = [probe() for _ in range(5)]
    stop = True
    th.join()
    print("%-18s %d frames in flight: %s MHz" % (name, a.n, " ".join("%.0f" % v for v in vals)))

import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ["JXL_PREPARE_TIMING"] = "1"
import numpy as np
from jxlatte_amd import _lib, host, synth
ctx = _lib.Context(0)
d = synth.make_vardct_frame(3840, 2160, seed=1000, mix="default")
for i in range(4):
    t0 = time.perf_counter()
    fr = host.Frame(ctx, d["params"], d["weights"], d["woffs"])
    t1 = time.perf_counter()
    for g in d["lfgroups"]:
        fr.setLFGroup(g)
    t2 = time.perf_counter()
    ctx.call("jxl_vardct_prepare")
    t3 = time.perf_counter()
    print("begin %.3f lfgroups %.3f prepare %.3f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3), flush=True)

"""The native-thread streaming leg of bench.py on its own (tools/native/stream_bench.cpp), for profiling:
    python tools/r5_stream.py [n_ctx] [frames_per_ctx]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from jxlatte_amd import _lib, abi, host, synth

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 8
fpc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
d = synth.make_vardct_frame(3840, 2160, seed=1000, mix="default")
p = abi.VarDCTParams.from_buffer_copy(d["params"])
p.transfer, p.out_format, p.stages = abi.TRANSFER_SRGB, abi.OUT_RGB8, 31
# the expected pixels: one frame through the synchronous calls
with _lib.Context(0) as c:
    fr = host.Frame(c, p, d["weights"], d["woffs"])
    for g in d["lfgroups"]:
        fr.setLFGroup(g)
    mp = fr.mapCoeffsI16()
    for ch in range(3):
        np.copyto(mp[ch], np.asarray(d["coeff"][ch], np.int16))
    fr.commitCoeffsI16()
    ref = fr.decodeFrame()
r = bench.streaming_leg_native(_lib, host, d, p, 0, 3840 * 2160, ref, n_ctx, fpc)
r.pop("note", None)
print(r)

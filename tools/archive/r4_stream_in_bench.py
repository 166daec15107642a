"""bench.py's own streaming leg, called (i) in a fresh process, (ii) after N steps of the 8-frame batch in the same process:
what in the bench process costs the leg 35 %?   python tools/r4_stream_in_bench.py [steps_before]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import bench
from jxlatte_amd import _lib, abi, host, synth
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 0
d = synth.make_vardct_frame(3840, 2160, seed=1000, mix="default")
p = abi.VarDCTParams.from_buffer_copy(d["params"])
p.transfer, p.out_format, p.stages = abi.TRANSFER_SRGB, abi.OUT_RGB8, 31
keep = []
if steps:
    ctxs = [_lib.Context(0) for _ in range(8)]
    frames = [host.Frame.from_synth(c, d) for c in ctxs]
    for _ in range(steps):
        for fr in frames:
            fr.run()
    for c in ctxs:
        c.synchronize()
    if os.environ.get("KEEP_CTX"):
        keep = [ctxs, frames]
    else:
        frames.clear()
        for c in ctxs:
            c.close()
if os.environ.get("NO_REF"):
    ref = np.zeros((2160, 3840, 3), np.uint8)
else:
    c = _lib.Context(0)
    fr = host.Frame(c, p, d["weights"], d["woffs"])
    for g in d["lfgroups"]:
        fr.setLFGroup(g)
    for grp in range(synth.num_groups(d)):
        fr.putGroup(0, grp, synth.group_view(d, grp))
    ref = fr.decodeFrame()
    if not os.environ.get("KEEP_REF_CTX"):
        c.close()
r = bench.streaming_leg(_lib, host, d, p, 0, 3840 * 2160, ref)
print("steps before %d keep %s:" % (steps, bool(keep)), r.get("streaming_end_to_end_Mpx_s"), r.get("ms_per_frame"), r.get("identical_output"), r.get("error"), r.get("host_ms_per_frame_and_thread"))

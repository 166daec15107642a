"""The streaming leg of bench.py on its own (N contexts, one host thread each, every frame through the whole boundary), for
profiling:  python tools/r4_stream.py [n_ctx] [frames_per_ctx]    -> wall time per frame; per-phase host times with PHASES=1"""
import ctypes as C, os, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
if os.environ.get("R4_STREAM_TORCH"):  # what does the bench process have that this one lacks? 1: torch imported, 2: + CUDA initialised + a sync
    import torch
    if os.environ["R4_STREAM_TORCH"] == "2":
        torch.cuda.set_device(0); torch.zeros(1, device="cuda"); torch.cuda.synchronize()
from jxlatte_amd import _lib, abi, host, synth

n_ctx = int(sys.argv[1]) if len(sys.argv) > 1 else 12
fpc = int(sys.argv[2]) if len(sys.argv) > 2 else 8
lib = _lib.load()
d = synth.make_vardct_frame(3840, 2160, seed=1000, mix="default")
p = abi.VarDCTParams.from_buffer_copy(d["params"])
p.transfer, p.out_format, p.stages = abi.TRANSFER_SRGB, abi.OUT_RGB8, 31
coeff16 = [np.ascontiguousarray(a, np.int16) for a in d["coeff"]]
all_groups = np.ones(synth.num_groups(d), np.uint8)
ctxs = [_lib.Context(0) for _ in range(n_ctx)]
pouts = [host.PinnedArray(lib, (2160, 3840, 3), np.uint8) for _ in range(n_ctx)]
start = threading.Barrier(n_ctx + 1)
t_end = [0.0] * n_ctx
phases = [np.zeros(8) for _ in range(n_ctx)]
PH = ["begin+weights", "lfgroups", "prepare", "map", "stores", "commit", "wait_prev+run", "read_begin"]


def worker(i):
    c = ctxs[i]
    pp = (C.c_void_p * 3)(pouts[i].array.ctypes.data, None, None)
    pending = [False]

    def one(acc):
        t = [time.perf_counter()]
        fr = host.Frame(c, p, d["weights"], d["woffs"]); t.append(time.perf_counter())
        for g in d["lfgroups"]:
            fr.setLFGroup(g)
        t.append(time.perf_counter())
        c.call("jxl_vardct_prepare"); t.append(time.perf_counter())
        mp = fr.mapCoeffsI16(no_fill=True); t.append(time.perf_counter())
        for ch in range(3):
            np.copyto(mp[ch], coeff16[ch])
        t.append(time.perf_counter())
        fr.commitCoeffsI16(all_groups); t.append(time.perf_counter())
        if pending[0]:
            c.call("jxl_vardct_read_output_wait")
        fr.run(); t.append(time.perf_counter())
        c.call("jxl_vardct_read_output_begin", pp, fr.width); t.append(time.perf_counter())
        pending[0] = True
        if acc is not None:
            acc += np.diff(t)
    one(None)
    c.call("jxl_vardct_read_output_wait"); pending[0] = False
    start.wait()
    for _ in range(fpc):
        one(phases[i])
    c.call("jxl_vardct_read_output_wait")
    t_end[i] = time.perf_counter()


th = [threading.Thread(target=worker, args=(i,)) for i in range(n_ctx)]
for t in th:
    t.start()
start.wait()
a = time.perf_counter()
for t in th:
    t.join()
wall = max(t_end) - a
n = n_ctx * fpc
print("%d contexts x %d frames: %.3f ms per frame wall = %.0f Mpx/s" % (n_ctx, fpc, wall * 1e3 / n, 3840 * 2160 * n / wall / 1e6))
tot = sum(phases) / n * 1e3
print("   per frame and thread, ms: " + ", ".join("%s %.2f" % (k, v) for k, v in zip(PH, tot)) + "  = %.2f" % tot.sum())

"""IDCT stage alone, per transform type (GPU box): one 4K frame tiled with a single varblock type (or a named mix), stage mask = IDCT
only, HIP-event time of the stage with the frame alone on the device.

    python tools/idct_types.py [--types DCT8,DCT16,...|all] [--mixes default,...] [--stamps out.npz] [--reps 20]

--stamps needs a -DJXL_STAMPS build of the library (JXL_AMD_LIB=...): lane 0 of every workgroup records s_memtime at its phase
boundaries; the per-phase medians and the launch timeline are printed and the raw array saved."""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np

from jxlatte_amd import _lib, abi, host, synth

ap = argparse.ArgumentParser()
ap.add_argument("--types", default="DCT8,DCT16,DCT32,DCT64,DCT16_8,DCT8_16,DCT32_8,DCT8_32,DCT32_16,DCT16_32,DCT4,DCT4_8,DCT2,HORNUSS,AFV0")
ap.add_argument("--mixes", default="default")
ap.add_argument("--reps", type=int, default=100)
ap.add_argument("--stamps", default="")
ap.add_argument("--size", default="3840x2160")
ap.add_argument("--stages", type=int, default=1)
args = ap.parse_args()
W, H = [int(v) for v in args.size.split("x")]
if args.stamps:
    import torch
    torch.zeros(1, device="cuda")  # torch brings its own HIP runtime: it has to initialise before the library does
ctx = _lib.Context(0)
lib = _lib.load()
print(lib.jxl_version().decode(), "lib:", _lib.SO_PATH)

stamp_buf = None
if args.stamps:
    import torch
    lib.jxl_debug_set_stamps.restype = C.c_int
    lib.jxl_debug_set_stamps.argtypes = [C.c_void_p]
    stamp_buf = torch.zeros((1 << 16, 8), dtype=torch.int64, device="cuda")
    assert lib.jxl_debug_set_stamps(C.c_void_p(stamp_buf.data_ptr())) == 0
    stamp3 = torch.zeros((1 << 13, 12), dtype=torch.int64, device="cuda")
    lib.jxl_debug_set_stamps3.restype = C.c_int
    lib.jxl_debug_set_stamps3.argtypes = [C.c_void_p]
    assert lib.jxl_debug_set_stamps3(C.c_void_p(stamp3.data_ptr())) == 0


def measure(frame, label):
    """steady state: the runs are enqueued back to back (the chip holds its clock; an idle chip between synchronised runs does
    not), the stage time is the mean of the last 32 runs' HIP events"""
    fr = host.Frame.from_synth(ctx, frame, stages=args.stages)
    for _ in range(args.reps):
        fr.run()
    ctx.call("jxl_vardct_enable_stage_timing", 1)
    for _ in range(32):
        fr.run()
    ctx.synchronize()
    v = C.c_float()
    ctx.call("jxl_vardct_last_stage_ms", 0, C.byref(v))
    ctx.call("jxl_vardct_enable_stage_timing", 0)
    us = v.value * 1e3
    # and alone on an idle chip (synchronised runs), for comparison
    ctx.call("jxl_vardct_enable_stage_timing", 1)
    for _ in range(8):
        fr.run()
        ctx.synchronize()
    ctx.call("jxl_vardct_last_stage_ms", 0, C.byref(v))
    ctx.call("jxl_vardct_enable_stage_timing", 0)
    px = frame["params"].width * frame["params"].height
    print("%-12s %8.1f us   %6.2f TB/s (24.3 B/px)   %6.1f Gpx/s   launches %d   (synchronised runs: %.1f us)"
          % (label, us, 24.3 * px / us / 1e6, px / us / 1e3, fr.lastLaunchCount(), v.value * 1e3), flush=True)
    return fr


names = [t[0] for t in abi.TRANSFORM_TYPES]
types = [t for t in args.types.split(",") if t]
if types == ["all"]:
    types = names
for t in types:
    frame = synth.make_vardct_frame(W, H, seed=1234, mix={t: 1.0})
    fr = measure(frame, t)
    if stamp_buf is not None:
        import torch
        stamp_buf.zero_()
        stamp3.zero_()
        torch.cuda.synchronize()
        fr.run()
        ctx.synchronize()
        torch.cuda.synchronize()
        s3 = stamp3.cpu().numpy()
        for which in (0, 1):
            rows = s3[which::2]
            rows = rows[rows[:, 8] != 0]
            if len(rows):
                names = ["dequant", "bar1", "prefetch", "colMAC", "bar2", "colwrite+bar3", "rowMAC", "stores", "llf", "bar4"]
                d = np.diff(rows[:, :11], axis=1).astype(np.float64)
                full = rows[:, 10] != 0
                print("   wg3 item %d of %d workgroups: " % (which, len(rows)) +
                      ", ".join("%s %d" % (nm, np.median(d[:, i])) for i, nm in enumerate(names[:8])) +
                      (", llf %d, bar4 %d" % (np.median(d[full, 8]), np.median(d[full, 9])) if full.any() else "") +
                      "; item span median %.0f" % np.median(rows[:, 8] - rows[:, 0]))
        st = stamp_buf.cpu().numpy()
        used = st[:, 0] != 0
        st = st[used]
        if len(st):
            t0 = st[:, 0].min()
            ph = np.diff(st[:, :6], axis=1).astype(np.float64)
            print("   workgroups %d; phase medians (cycles): %s; total median %.0f, p90 %.0f; start spread: p50 %.0f p90 %.0f max %.0f; kernel span %.0f cycles"
                  % (len(st), np.median(ph, axis=0).astype(int).tolist(), np.median(st[:, 5] - st[:, 0]), np.percentile(st[:, 5] - st[:, 0], 90),
                     np.median(st[:, 0] - t0), np.percentile(st[:, 0] - t0, 90), (st[:, 0] - t0).max(), st[:, 5].max() - t0))
            os.makedirs(os.path.dirname(args.stamps) or ".", exist_ok=True)
            np.savez_compressed(args.stamps.replace(".npz", "_%s.npz" % t), stamps=st)
for m in [m for m in args.mixes.split(",") if m]:
    # a named mix, or an explicit one: DCT8=0.5+DCT16=0.5
    mix = {kv.split("=")[0]: float(kv.split("=")[1]) for kv in m.split("+")} if "=" in m else m
    frame = synth.make_vardct_frame(W, H, seed=1234, mix=mix)
    fr = measure(frame, "mix:" + m)
    if stamp_buf is not None:
        import torch
        stamp3.zero_()
        torch.cuda.synchronize()
        fr.run()
        ctx.synchronize()
        torch.cuda.synchronize()
        s3 = stamp3.cpu().numpy().astype(np.uint64)
        t_in, t_out = s3[0::2, 11], s3[1::2, 11]
        ok = t_in != 0
        t_in, cnt, t_out = t_in[ok].astype(np.int64), (t_out[ok] >> np.uint64(56)).astype(np.int64), (t_out[ok] & np.uint64((1 << 56) - 1)).astype(np.int64)
        t0 = t_in.min()
        life = t_out - t_in
        print("   wg3 lifetimes of %d workgroups (cycles of s_memtime): entry p50 %d p90 %d max %d | life p10 %d p50 %d p90 %d max %d | exit p10 %d p50 %d p90 %d max %d | items %d..%d"
              % (len(t_in), np.median(t_in - t0), np.percentile(t_in - t0, 90), (t_in - t0).max(), np.percentile(life, 10), np.median(life),
                 np.percentile(life, 90), life.max(), np.percentile(t_out - t0, 10), np.median(t_out - t0), np.percentile(t_out - t0, 90),
                 (t_out - t0).max(), cnt.min(), cnt.max()))
        for x in range(8):
            sel = (np.nonzero(ok)[0] % 8) == x
            print("      XCD %d: life p50 %d max %d, exit max %d" % (x, np.median(life[sel]), life[sel].max(), (t_out[sel] - t0).max()))

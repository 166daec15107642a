#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the jxlatte transform stage on MI355X.

    python bench.py --gpus N --steps K --warmup W

One process per GPU (torch.distributed over RCCL when N > 1; RANK/LOCAL_RANK/WORLD_SIZE/MASTER_* from
the environment). A *step* = one pass of the hot path (dequant + CfL + IDCT -> Gab -> EPF -> XYB) over
this rank's batch of independent synthetic 4K VarDCT frames (workload C3/C5:
seeds 1000+..., 8 frames per GPU by default), inputs already resident in HBM. Frames are independent,
so the path shards by frame with NO data-path collective inside a step ("scaling": "weak"); the RCCL
gather of finished pixels to rank 0 (north_star's "trivial gather") is timed separately and reported
under "gather". Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="vardct4k", choices=["vardct4k", "vardct8k_pq", "modular1080p", "modular8k", "jxlfile"])
    ap.add_argument("--input", default=os.path.join(os.path.dirname(os.path.abspath(__file__)), "tests", "golden", "samples", "bbb.jxl"),
                    help="--workload jxlfile: a VarDCT .jxl file parsed by the C++ front-end (real varblock statistics)")
    ap.add_argument("--frames-per-gpu", type=int, default=8)
    ap.add_argument("--distinct-frames", type=int, default=2, help="distinct synthetic frames generated per rank (the rest reuse them)")
    ap.add_argument("--mix", default="default")
    ap.add_argument("--epf-iters", type=int, default=2)
    ap.add_argument("--streams", type=int, default=0, help="0 = one HIP stream per frame context (default); 1 = all frames of a rank share one stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--batch", action="store_true", help="jxl_vardct_run_batch instead of one jxl_vardct_run per frame")
    ap.add_argument("--stages", type=int, default=31, help="stage mask (diagnostics): 1 IDCT, 2 Gab, 4 EPF, 8 XYB, 16 out")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) and run the collective legs even with one rank")
    ap.add_argument("--verify", action="store_true", help="check frame 0 against the oracle before timing")
    return ap.parse_args()


_REAL_STDOUT = None


def emit(obj):
    """the ONE JSON line, on the process's real stdout"""
    data = (json.dumps(obj) + "\n").encode()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, data)


def main():
    global _REAL_STDOUT
    args = parse()
    # RCCL prints a version banner on stdout when the first communicator is created; keep stdout clean for the
    # single JSON line by pointing fd 1 at stderr for the rest of the run
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from jxlatte_amd import _lib, abi, host, synth

    if args.workload.startswith("modular"):
        return bench_modular(args, rank, world, local_rank, torch, dist)

    W, H = (3840, 2160) if args.workload == "vardct4k" else (7680, 4320)
    fpg = args.frames_per_gpu
    real_stats = None
    kw = dict(epf_iters=args.epf_iters)
    if args.workload == "vardct8k_pq":
        kw.update(transfer=abi.TRANSFER_PQ, out_format=abi.OUT_U16, opsin_matrix=synth.bt2100_opsin_matrix(), intensity_target=10000.0)
        fpg = min(fpg, 2)
    # ---- synthetic inputs -> HBM (untimed)
    t0 = time.time()
    distinct = []
    for i in range(0 if args.workload == "jxlfile" else min(args.distinct_frames, fpg)):
        seed = 1234 if (world == 1 and fpg == 1) else 1000 + rank * fpg + i
        distinct.append(synth.make_vardct_frame(W, H, seed=seed, mix=args.mix, **kw))
    ctxs, frames = [], []
    for i in range(fpg):
        c = _lib.Context(local_rank)
        if args.streams == 1 and ctxs:
            c.call("jxl_ctx_set_stream", ctxs[0].stream)
        ctxs.append(c)
        if args.workload == "jxlfile":
            from jxlatte_amd.decoder import load_vardct_frame
            fr_, real_stats = load_vardct_frame(args.input, c)
            W, H = real_stats["padded_width"], real_stats["padded_height"]
            frames.append(fr_)
        else:
            frames.append(host.Frame.from_synth(c, distinct[i % len(distinct)], stages=args.stages))
    gen_s = time.time() - t0
    npx = W * H
    lib = _lib.load()

    if args.verify and rank == 0 and distinct:
        from oracle import pyoracle as orc
        got = frames[0].decodeFrame()
        exp = orc.vardct_frame(distinct[0], threads=os.cpu_count())
        if got.dtype == np.float32:
            ok = np.array_equal(got.view(np.uint32), exp.view(np.uint32))
        else:
            ok = np.abs(got.astype(np.int64) - exp.astype(np.int64)).max() <= 1
        print("verify frame 0 vs oracle:", "OK" if ok else "MISMATCH", file=sys.stderr)
        if not ok:
            raise SystemExit(2)

    def step():
        # one jxl_vardct_run per frame, each on its own stream (the faster form); --batch: the batch entry of the
        # C-ABI (IDCT stage of all frames in shared launches: 12 launches per 8 frames instead of 40, but 5 % slower)
        if not args.batch or len(frames) == 1:
            for fr in frames:
                fr.run()
        else:
            host.Frame.runBatch(frames)

    def sync_all():
        torch.cuda.synchronize()
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    ctxs[0].call("jxl_vardct_enable_stage_timing", 1)
    t1 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    dt = torch.tensor([t2 - t1], dtype=torch.float64, device="cuda")
    if use_dist:
        dist.barrier()
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    torch.cuda.synchronize()
    elapsed = float(dt.item())
    ms_per_step = elapsed * 1e3 / args.steps
    total_px = float(npx) * fpg * world * args.steps
    value = total_px / elapsed / 1e6  # Mpixels/s, whole job

    # HIP-event stage times of ctx 0 (averaged over its runs inside the timed region)
    def stage_ms():
        out = []
        for which in (0, 1, 2):
            v = C.c_float()
            ctxs[0].call("jxl_vardct_last_stage_ms", which, C.byref(v))
            out.append(v.value)
        return out
    ms_all_b, ms_idct_b, ms_rest_b = stage_ms()  # inside the timed region (other frames' kernels overlap)
    launches = frames[0].lastLaunchCount()
    # the same events with frame 0 alone on the device: the kernels' own durations (what rocprofv3 reports per launch)
    torch.cuda.synchronize()
    ctxs[0].call("jxl_vardct_enable_stage_timing", 1)
    for _ in range(max(5, min(args.steps, 30))):
        frames[0].run()
        ctxs[0].synchronize()
    ms_all, ms_idct, ms_rest = stage_ms()
    ctxs[0].call("jxl_vardct_enable_stage_timing", 0)

    # single-frame latency (one frame alone on the device)
    lat = []
    for _ in range(5):
        torch.cuda.synchronize()
        a = time.perf_counter()
        frames[0].run()
        ctxs[0].synchronize()
        lat.append((time.perf_counter() - a) * 1e3)
    single_ms = float(np.median(lat))

    # ---- optional RCCL gather of the finished pixels to rank 0 (timed on its own)
    gather = None
    if use_dist and not args.no_gather:
      try:
        es = lib.jxl_vardct_out_elem_size(ctxs[0].h)
        nbytes = 3 * npx * es * fpg
        send = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        recv = [torch.empty(nbytes, dtype=torch.uint8, device="cuda") for _ in range(world)] if rank == 0 else None
        for i, c in enumerate(ctxs):
            c.call("jxl_vardct_copy_output_device", C.c_void_p(send.data_ptr() + i * 3 * npx * es))
        sync_all()
        g0 = time.perf_counter()
        for _ in range(3):
            dist.gather(send, recv, dst=0)
        torch.cuda.synchronize()
        gdt = torch.tensor([(time.perf_counter() - g0) / 3], dtype=torch.float64, device="cuda")
        dist.all_reduce(gdt, op=dist.ReduceOp.MAX)
        gather = {"ms_per_step": round(float(gdt.item()) * 1e3, 3), "payload_MB_per_rank": round(nbytes / 1e6, 1),
                  "note": "ncclGather of one step's output planes to rank 0; not inside the timed steps"}
      except Exception as e:  # the optional leg must never take the measurement down
        gather = {"error": repr(e)[:200]}

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- roofline of the dominant kernel (restoration + colour stage) and of the whole path
    out_bytes_px = 6.0 if args.workload == "vardct8k_pq" else 12.0
    side = 21.0 / 64.0  # per-pixel share of the 8x8-cell side info
    path_bytes = (12.0 + out_bytes_px + side) * npx + 1.58e6  # whole path, algorithmic (24.5 B/px for f32 out)
    rest_bytes = (12.0 + out_bytes_px + 8.0 / 64.0) * npx     # restore stage: planes in + planes out + hfMul/sharpness
    rest_s = ms_rest * 1e-3
    achieved = rest_bytes / rest_s / 1e9 if rest_s > 0 else 0.0
    path_gbs = path_bytes * fpg * args.steps / elapsed / 1e9
    # HBM bytes per launch of this kernel from the committed PMC passes of the same configuration (profiles/):
    # rocprofv3 cannot run inside the bench, so the figure is carried over when the workload matches
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "r1_traffic.json")
    if args.workload == "vardct4k" and args.epf_iters == 2 and os.path.exists(tpath):
        try:
            traffic = int(json.load(open(tpath))["hbm_bytes_per_launch"])
        except Exception:
            traffic = None
    roofline = {
        "bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic,
        "kernel": "k_restore_fused (Gab, EPF x%d, XYB): HIP events around the launch, frame 0 alone on the device" % args.epf_iters,
        "kernel_ms": round(ms_rest, 4), "algorithmic_bytes_per_launch": int(rest_bytes),
        "kernel_ms_in_batch": round(ms_rest_b, 4),
        "idct_stage_ms": round(ms_idct, 4), "idct_stage_ms_in_batch": round(ms_idct_b, 4), "frame_ms_events": round(ms_all, 4),
        "path_algorithmic_GBps": round(path_gbs, 1), "path_frac": round(path_gbs / HBM_PEAK_GBS, 4),
        "note": "this kernel is bound by VALU issue and LDS, not HBM: 533 non-fusable f32 instructions per output pixel in the reference's summation order = 73 us at the measured issue peak, 43 us of LDS time, 25 us of HBM time (DESIGN.md 4.2)",
    }

    cpu = None
    if not args.no_cpu_baseline and distinct:
        # bounded sample (~25 core-seconds of CPU work): the frame of the same workload 3 times on all host cores, and
        # once on ONE core (the reference's transform stage is single-threaded)
        from oracle import pyoracle as orc
        ncores = min(os.cpu_count() or 1, 64)
        fr0 = distinct[0]
        reps = 3
        orc.vardct_frame(fr0, threads=ncores)  # page in
        a = time.perf_counter()
        for _ in range(reps):
            orc.vardct_frame(fr0, threads=ncores)
        t_all = (time.perf_counter() - a) / reps
        a = time.perf_counter()
        orc.vardct_frame(fr0, threads=1)
        t_1 = time.perf_counter() - a
        cpu = {"value": round(npx / t_all / 1e6, 2), "unit": "Mpixels/s", "cores": ncores, "kind": "port",
               "sample": "%d x 1 frame %dx%d of the same workload on %d OpenMP threads (mean); 1-core figure from the same frame once. "
                         "C oracle = line-faithful restatement of the Java path (no JVM on this box)" % (reps, W, H, ncores),
               "seconds": round(t_all * reps, 2),
               "value_1core": round(npx / t_1 / 1e6, 3), "seconds_1core": round(t_1, 2)}

    line = {
        "metric": "Mpixels/s decoded (VarDCT 4K frame) at 1/2/4/8 MI355X vs host-CPU jxlatte",
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" if distinct else "real bitstream %s (parsed by the C++ front-end)" % os.path.basename(args.input),
        "config": {"workload": "%s: %d independent %dx%d VarDCT frames per GPU (mix=%s, Gab + EPF x%d + XYB, %s out), inputs resident in HBM"
                               % (args.workload, fpg, W, H, args.mix if distinct else "as coded", real_stats["epf_iters"] if real_stats else args.epf_iters,
                                  "f32" if out_bytes_px == 12.0 else "PQ u16"),
                   "frames_per_gpu": fpg, "distinct_frames": len(distinct) or 1, "streams": args.streams if args.streams else fpg,
                   "varblock_area_share": synth.type_histogram(distinct[0]) if distinct else real_stats["varblocks"],
                   "kernel_launches_per_frame": launches,
                   "single_frame_ms": round(single_ms, 4),
                   "single_frame_Mpx_s": round(npx / single_ms / 1e3, 1), "input_gen_s": round(gen_s, 1)},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    if gather:
        line["gather"] = gather
    emit(line)
    if use_dist:
        dist.destroy_process_group()


def bench_modular(args, rank, world, local_rank, torch, dist):
    from jxlatte_amd import _lib, host, synth
    W, H = (1920, 1080) if args.workload == "modular1080p" else (7680, 4320)
    fpg = max(1, min(args.frames_per_gpu, 4))
    streams, ctxs = [], []
    mod = synth.make_modular_frame(W, H, channels=3, seed=7 + rank)
    for i in range(fpg):
        c = _lib.Context(local_rank)
        if args.streams == 1 and ctxs:
            c.call("jxl_ctx_set_stream", ctxs[0].stream)
        ms = host.ModularStream(c, mod["chans"], mod["sp"])
        ms.begin()
        ctxs.append(c)
        streams.append(ms)

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        for ms in streams:
            ms.run()
    sync_all()
    t1 = time.perf_counter()
    for _ in range(args.steps):
        for ms in streams:
            ms.run()
    torch.cuda.synchronize()
    dt = torch.tensor([time.perf_counter() - t1], dtype=torch.float64, device="cuda")
    if world > 1:
        dist.barrier()
        dist.all_reduce(dt, op=dist.ReduceOp.MAX)
    elapsed = float(dt.item())
    if rank != 0:
        return
    npx = W * H
    value = npx * fpg * world * args.steps / elapsed / 1e6
    bytes_img = 24.0 * npx
    gbs = bytes_img * fpg * args.steps / elapsed / 1e9
    cpu = None
    if not args.no_cpu_baseline:
        from oracle import pyoracle as orc
        orc.modular_apply(mod["chans"], mod["sp"])  # page in
        reps, a = 0, time.perf_counter()
        while reps < 40 and time.perf_counter() - a < 2.0:  # bounded sample: about 2 s of wall time on all host cores
            orc.modular_apply(mod["chans"], mod["sp"])
            reps += 1
        t = (time.perf_counter() - a) / reps
        cpu = {"value": round(npx / t / 1e6, 2), "unit": "Mpixels/s", "cores": os.cpu_count(), "kind": "port",
               "sample": "%d x 1 image %dx%dx3 (mean), C oracle (H steps OpenMP over rows)" % (reps, W, H), "seconds": round(t * reps, 3)}
    emit({
        "metric": "Mpixels/s inverse Squeeze (Modular %dx%d, 3 channels, default squeeze plan)" % (W, H),
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": "%s: %d images per GPU, %d squeeze steps" % (args.workload, fpg, len(mod["sp"])),
                   "launches": ctxs[0].lib.jxl_modular_last_launch_count(ctxs[0].h)},
        "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": None,
                     "note": "whole step list (24 B/px algorithmic over the elapsed time of all steps); segmented walk of the serial squeeze recurrence: 64-pair segments with a 16-pair warm-up, verified and redone serially where a boundary state differs (DESIGN.md 4.3)"},
        "cpu_baseline": cpu,
    })


if __name__ == "__main__":
    main()

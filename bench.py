#!/usr/bin/env python3
"""bench.py -- Mpixels/s of the jxlatte transform stage on MI355X.

    python bench.py --gpus N --steps K --warmup W

One process per GPU. With `--gpus N > 1` and no WORLD_SIZE in the environment, this process spawns the N
ranks itself (fresh child processes, before anything touches the GPU) and forwards rank 0's JSON line;
under `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N` it IS one of the ranks
(RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment). torch.distributed over RCCL carries the
barrier, the max-over-ranks time and the gather.

A *step* = one pass of the hot path (dequant + CfL + IDCT -> Gab -> EPF -> XYB) over this rank's share of a
batch of independent synthetic 4K VarDCT frames (workload C3/C5: frame i has seed
1000 + i and goes to rank i mod N -- jxlatte_amd.shard.frames_of_rank -- 8 frames per GPU by default), inputs
already resident in HBM. Frames are independent: no data-path collective inside a step ("scaling": "weak").
The RCCL gather of finished pixels to rank 0 (north_star's "trivial gather", shard.gather_planes) is reported
under "gather": on its own and overlapped with the next step's compute, for f32 planes and for RGB16 output.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import math
import os
import shutil
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling
# non-fused f32 VALU issue peak: 1024 SIMDs x one wave-instruction (64 lanes) per 2 cycles at the 2.4 GHz maximum clock
# (MI355X_MICROARCH.md "Per-instruction cycle constants": v_add_f32 / v_mul_f32 2 cyc per wave64 on a SIMD-32)
VALU_PEAK_GINST = 1024 * 2.4 / 2.0  # 1228.8 G wave-instructions / s


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="vardct4k", choices=["vardct4k", "vardct8k_pq", "modular1080p", "modular8k", "jxlfile"])
    ap.add_argument("--input", default=os.path.join(ROOT, "tests", "golden", "samples", "bbb.jxl"),
                    help="--workload jxlfile: a VarDCT .jxl file parsed by the C++ front-end (real varblock statistics)")
    ap.add_argument("--frames-per-gpu", type=int, default=8)
    ap.add_argument("--distinct-frames", type=int, default=8, help="distinct synthetic frames generated per rank (the rest reuse them); default: every frame of the C5 share its own seed 1000 + i (SURVEY 8(d))")
    ap.add_argument("--mix", default="default")
    ap.add_argument("--stream-groups", type=int, default=2,
                    help="the frames' contexts share G main streams (jxl_ctx_set_stream): frames i, i + G, ... are enqueued behind one another, G frames are "
                         "in flight (0: one stream per frame -- the default until r5). r6: with the IDCT stage as ONE launch per frame, two frames in flight "
                         "are 4 % faster than eight (profiles/experiments/r6_stream_groups_prio.txt)")
    ap.add_argument("--size", default="", help="WxH override of the synthetic VarDCT frame size (diagnostics; named in config.workload)")
    ap.add_argument("--epf-iters", type=int, default=2)
    ap.add_argument("--streams", type=int, default=0, help="0 = one HIP stream per frame context (default); 1 = all frames of a rank share one stream")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-gather", action="store_true")
    ap.add_argument("--no-end-to-end", action="store_true", help="skip the host_prepare / H2D / D2H / end-to-end leg")
    ap.add_argument("--batch", action="store_true", help="jxl_vardct_run_batch instead of one jxl_vardct_run per frame")
    ap.add_argument("--stages", type=int, default=31, help="stage mask (diagnostics): 1 IDCT, 2 Gab, 4 EPF, 8 XYB, 16 out")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed (RCCL) and run the collective legs even with one rank")
    ap.add_argument("--verify", action="store_true", help="check frame 0 against the oracle before timing")
    ap.add_argument("--no-also", action="store_true", help="skip the `also` object (the other workloads north_star names, each run as a child process after the headline)")
    ap.add_argument("--streaming-child", default="", help="internal: run the streaming boundary leg alone (device,contexts,frames_per_context,job.pkl) and print its JSON")
    return ap.parse_args(argv)


_REAL_STDOUT = None


def emit(obj):
    """the ONE JSON line, on the process's real stdout"""
    data = (json.dumps(obj) + "\n").encode()
    os.write(_REAL_STDOUT if _REAL_STDOUT is not None else 1, data)


# ---------------------------------------------------------------------------------------------------------------
def spawn_ranks(args):
    """--gpus N without a launcher: start N fresh rank processes (this process never touches the GPU: counting devices
    does not initialise HIP on this image, everything else happens in the children), wait, fail if any rank fails."""
    import socket
    import torch
    n = args.gpus
    have = torch.cuda.device_count()
    if have < n:
        sys.stderr.write("bench.py --gpus %d: only %d GPU(s) visible on this node; refusing to oversubscribe "
                         "(a rank per GPU is the contract)\n" % (n, have))
        return 2
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        # rank 0 inherits stdout (its JSON line is this command's output); the others must not write there
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    pending = set(range(n))
    while pending:
        for r in list(pending):
            code = procs[r].poll()
            if code is None:
                continue
            pending.discard(r)
            if code != 0 and rc == 0:
                rc = code if code > 0 else 1
                sys.stderr.write("bench.py: rank %d exited with %d; stopping the other ranks\n" % (r, code))
                for q in pending:
                    procs[q].terminate()
        time.sleep(0.05)
    return rc


ALSO_RUNS = (
    # key, extra arguments, what it is
    ("modular8k", ["--workload", "modular8k", "--frames-per-gpu", "1"], "8K Modular (7680x4320x3, default squeeze plan), one image alone"),
    ("modular8k_x4", ["--workload", "modular8k", "--frames-per-gpu", "4", "--no-cpu-baseline"], "the same, four images in flight"),
    ("modular1080p", ["--workload", "modular1080p", "--frames-per-gpu", "1"], "config C2: one 1080p Modular image alone"),
    ("vardct8k_pq", ["--workload", "vardct8k_pq", "--no-cpu-baseline"], "config C4: 8K VarDCT, XYB -> linear -> PQ u16 (2 frames per step)"),
    ("vardct4k_epf1", ["--epf-iters", "1", "--no-cpu-baseline"], "the headline workload with 1 EPF iteration (SURVEY 8(d))"),
    ("vardct4k_epf3", ["--epf-iters", "3", "--no-cpu-baseline"], "the headline workload with 3 EPF iterations (SURVEY 8(d))"),
)


def also_runs(args):
    """The other workloads north_star names -- 8K Modular (1 and 4 in flight), config C2 (1080p Modular), config C4 (8K PQ), the 1- and
    3-iteration EPF runs -- measured by THIS command so that the driver's line carries them: each is this script
    as a child process (fresh contexts, same timing rules: warm-up, then K steps between synchronisations), started only after the
    headline's timed region and its other legs are over, so nothing shares the device with `value`. Never folded into `value`."""
    out = {}
    t0 = time.time()
    for key, extra, what in ALSO_RUNS:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--no-gather", "--no-end-to-end", "--no-also"] + extra
        env = dict(os.environ)
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "LOCAL_WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
            env.pop(k, None)
        a = time.time()
        try:
            r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
            js = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode != 0 or not js:
                out[key] = {"what": what, "error": "exit %d: %s" % (r.returncode, (r.stderr or "")[-300:])}
                continue
            d = json.loads(js[-1])
        except Exception as ex:  # a failed side run must not cost the headline its line
            out[key] = {"what": what, "error": repr(ex)[:300]}
            continue
        rf = d.get("roofline") or {}
        e = {"what": what, "value": d.get("value"), "unit": d.get("unit"), "ms_per_step": d.get("ms_per_step"), "dtype": d.get("dtype"),
             "workload": (d.get("config") or {}).get("workload"),
             "frac": rf.get("path_frac") if rf.get("path_frac") is not None else rf.get("frac"),
             "frac_of": "whole path, algorithmic bytes / elapsed / 8 TB/s" if rf.get("path_frac") is not None else "algorithmic bytes (24 B/px) / elapsed / 8 TB/s",
             "kernel_frac": rf.get("frac") if rf.get("path_frac") is not None else None,
             "traffic": rf.get("traffic"), "traffic_source": rf.get("traffic_source"), "run_s": round(time.time() - a, 1)}
        if (d.get("config") or {}).get("single_frame_ms") is not None:
            e["single_frame_ms"] = d["config"]["single_frame_ms"]
        if d.get("cpu_baseline"):
            e["cpu_baseline"] = d["cpu_baseline"]
        out[key] = e
    out["note"] = ("each entry: `python bench.py <arguments of ALSO_RUNS>` as a child process after the headline's legs, %d steps after %d warm-up "
                   "steps; `traffic` = HBM bytes per launch / plan from profiles/*traffic.json while the kernel sources hash to what was "
                   "profiled, else null; never part of `value`; %.0f s in all" % (args.steps, args.warmup, time.time() - t0))
    return out



def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def jvm_probe(sample_path):
    """SURVEY 8(d) / BASELINE.md 3.4: the true reference is Java. If a JVM and a user-supplied JXLATTE_JAR exist, time
    `java -jar $JXLATTE_JAR in.jxl out.png` on a real file for context; otherwise say so."""
    java = shutil.which("java")
    jar = os.environ.get("JXLATTE_JAR")
    if not java or not jar or not os.path.exists(jar):
        msg = "JVM: absent" if not java else "JVM: present, JXLATTE_JAR not set" if not jar else "JVM: present, %s missing" % jar
        print(msg + " (the Java reference cannot be timed on this box; cpu_baseline is the C restatement)", file=sys.stderr)
        return {"jvm": msg}
    try:
        ver = subprocess.run([java, "-version"], capture_output=True, text=True, timeout=30).stderr.splitlines()[0]
        out = os.path.join("/tmp", "jxlatte_ref_%d.png" % os.getpid())
        best = None
        for _ in range(2):  # second run: page cache + JIT-warm class data sharing
            a = time.perf_counter()
            r = subprocess.run([java, "-jar", jar, sample_path, out], capture_output=True, timeout=600)
            dt = time.perf_counter() - a
            if r.returncode != 0:
                return {"jvm": ver, "error": r.stderr.decode("utf-8", "replace")[-200:]}
            best = dt if best is None else min(best, dt)
        return {"jvm": ver, "jar": os.path.basename(jar), "file": os.path.basename(sample_path), "decode_to_png_s": round(best, 3),
                "note": "whole java -jar invocation (JVM start, entropy decode, transforms, PNG encode), not the transform stage alone"}
    except Exception as e:  # the probe must never take the measurement down
        return {"jvm": "probe failed: %r" % (e,)}


# ---------------------------------------------------------------------------------------------------------------
def streaming_child(spec):
    """the streaming boundary leg in a process of its own (bench.py's parent passes device, contexts, frames per context and the
    expected pixels): no torch, no other contexts or streams of the timed legs, and the runtime setting a multi-decoder host
    runs with (GPU_MAX_HW_QUEUES, set by the parent before this process initialises HIP)"""
    import pickle
    dev, n_ctx, fpc, path = spec.split(",", 3)
    from jxlatte_amd import _lib, abi, host
    with open(path, "rb") as f:
        job = pickle.load(f)  # the parent's frame, its parameter block and the pixels it got for it
    p = abi.VarDCTParams.from_buffer_copy(job["params"])
    r = streaming_leg(_lib, host, job["frame"], p, int(dev), job["npx"], job["ref"], int(n_ctx), int(fpc), in_child=True)
    print(json.dumps(r))


def main():
    global _REAL_STDOUT
    args = parse()
    if args.streaming_child:
        return streaming_child(args.streaming_child)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))
    # RCCL prints a version banner on stdout when the first communicator is created; keep stdout clean for the
    # single JSON line by pointing fd 1 at stderr for the rest of the run
    sys.stdout.flush()
    _REAL_STDOUT = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != max(args.gpus, 1) and rank == 0:
        print("bench.py: WORLD_SIZE=%d but --gpus %d; the launcher's world size is used" % (world, args.gpus), file=sys.stderr)
    import torch
    import torch.distributed as dist
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    if local_rank >= torch.cuda.device_count():
        raise SystemExit("bench.py: rank %d has no GPU of its own (%d visible)" % (rank, torch.cuda.device_count()))
    torch.cuda.set_device(local_rank)
    use_dist = world > 1 or args.force_dist
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    from jxlatte_amd import _lib, abi, host, shard, synth

    if args.workload.startswith("modular"):
        return bench_modular(args, rank, world, local_rank, torch, dist)

    W, H = (3840, 2160) if args.workload == "vardct4k" else (7680, 4320)
    if args.size:
        W, H = (int(v) for v in args.size.lower().split("x"))
    fpg = args.frames_per_gpu
    real_stats = None
    kw = dict(epf_iters=args.epf_iters)
    if args.workload == "vardct8k_pq":
        kw.update(transfer=abi.TRANSFER_PQ, out_format=abi.OUT_U16, opsin_matrix=synth.bt2100_opsin_matrix(), intensity_target=10000.0)
        fpg = min(fpg, 2)
    # ---- the batch: n_frames = fpg * world independent frames, frame i on rank i mod world (shard.frames_of_rank);
    #      synthetic inputs -> HBM (untimed). Frame i has seed 1000 + i (C5); a lone frame uses the C3 seed 1234.
    n_frames = fpg * world
    my_frames = shard.frames_of_rank(n_frames, rank, world)
    t0 = time.time()
    distinct = []
    for j in range(0 if args.workload == "jxlfile" else min(args.distinct_frames, fpg)):
        seed = 1234 if n_frames == 1 else 1000 + my_frames[j]
        distinct.append(synth.make_vardct_frame(W, H, seed=seed, mix=args.mix, **kw))
    ctxs, frames = [], []
    for i in range(fpg):
        c = _lib.Context(local_rank)
        if args.streams == 1 and ctxs:
            c.call("jxl_ctx_set_stream", ctxs[0].stream)
        elif args.stream_groups and i >= args.stream_groups:  # contexts i, i + G, ... launch on ONE main stream
            c.call("jxl_ctx_set_stream", ctxs[i % args.stream_groups].stream)
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
        # C-ABI (IDCT stage of all frames in shared launches)
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

    def timed_region():
        """EXACTLY args.steps steps between barrier + synchronize on both sides; max over ranks"""
        sync_all()
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
        return float(dt.item())

    # One timed region of K steps of a few milliseconds is one thin sample (28 ms, +-3 % between variants is inside its
    # noise): the region is repeated -- at least 5 times and until 0.25 s of timed work have accumulated (the count is agreed on
    # between the ranks) -- and `value` / `ms_per_step` are the MEDIAN repetition's; min / max / spread are reported beside it.
    first = timed_region()
    n_rep = max(5, min(400, int(math.ceil(0.25 / max(first, 1e-6)))))
    if use_dist:
        nr = torch.tensor([n_rep], dtype=torch.int64, device="cuda")
        dist.all_reduce(nr, op=dist.ReduceOp.MAX)
        n_rep = int(nr.item())
    reps_s = [first] + [timed_region() for _ in range(n_rep - 1)]
    elapsed = float(np.median(reps_s))
    timing = {"repetitions": len(reps_s), "timed_s_total": round(float(np.sum(reps_s)), 4),
              "ms_per_step_min": round(min(reps_s) * 1e3 / args.steps, 4), "ms_per_step_median": round(elapsed * 1e3 / args.steps, 4),
              "ms_per_step_max": round(max(reps_s) * 1e3 / args.steps, 4),
              "spread_pct": round(100.0 * (max(reps_s) - min(reps_s)) / elapsed, 2),
              "note": "each repetition = exactly `steps` steps between barrier + synchronize; value and ms_per_step are the median repetition"}
    ms_per_step = elapsed * 1e3 / args.steps
    total_px = float(npx) * n_frames * args.steps
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
    # (r4: the runs are enqueued back to back -- the frame's launches are ordered on its streams, so each run still has the device
    # to itself, but the chip holds its clock; with a host synchronisation between the runs it does not, and the figures of one
    # box ranged over 104-129 us for the same kernels)
    torch.cuda.synchronize()
    for _ in range(8):
        frames[0].run()
    ctxs[0].call("jxl_vardct_enable_stage_timing", 1)
    for _ in range(32):
        frames[0].run()
    ctxs[0].synchronize()
    ms_all, ms_idct, ms_rest_stage = stage_ms()
    # r6: the dominant kernel's OWN start -> stop (events recorded by the launch itself: what rocprofv3 reports for it). The stage's
    # stream events (ms_rest_stage) also hold the boundary between the IDCT launch and this one -- since the IDCT stage is ONE launch
    # per frame nothing sits between its end and the stage event any more, and the two figures differ by that boundary (~10 us)
    ms_rest = ms_rest_stage
    try:
        v = C.c_float()
        ctxs[0].call("jxl_vardct_last_stage_ms", 3, C.byref(v))
        if v.value > 0:
            ms_rest = v.value
    except Exception:
        pass
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

    # ---- RCCL gather of the finished pixels to rank 0 through shard.gather_planes (what tests/test_shard_cpu.py covers with
    #      gloo): (i) on its own, (ii) overlapped with the next step's compute (double-buffered send tensors: the gather of
    #      step k runs on RCCL's stream while the ctx streams compute step k+1) -- SURVEY 8(e)'s two numbers
    gather = None
    if use_dist and not args.no_gather:
        try:
            gather = {"f32": gather_legs(args, torch, dist, shard, lib, ctxs, frames, step, sync_all, n_frames, rank, world, H, W)}
            if args.workload == "vardct4k" and distinct:
                # the same batch with the device output stage on (sRGB transfer + RGB16 interleave, row f3): half the bytes
                ctx2, fr2 = [], []
                for i in range(fpg):
                    c = _lib.Context(local_rank)
                    d = dict(distinct[i % len(distinct)])
                    p = abi.VarDCTParams.from_buffer_copy(d["params"])
                    p.transfer, p.out_format = abi.TRANSFER_SRGB, abi.OUT_RGB16
                    d["params"] = bytes(p)
                    ctx2.append(c)
                    fr2.append(host.Frame.from_synth(c, d, stages=31))

                def step2():
                    for fr in fr2:
                        fr.run()
                gather["rgb16"] = gather_legs(args, torch, dist, shard, lib, ctx2, fr2, step2, sync_all, n_frames, rank, world, H, W)
                for c in ctx2:
                    c.close()
        except Exception as e:  # the optional leg must never take the measurement down
            gather = {"error": repr(e)[:300]}

    if rank != 0:
        if use_dist:
            dist.destroy_process_group()
        return

    # ---- what the timed step leaves out (one 4K frame, rank 0): host preparation, H2D, D2H, end to end with RGB8 output
    e2e = None
    if not args.no_end_to_end and distinct and args.workload == "vardct4k":
        try:
            # the timed batch is over and every figure of it has been read: its contexts (8 frames' planes, 16+ streams on the
            # process's 4 hardware queues) go before the boundary legs create their own -- with them alive the streaming leg
            # measured 3.0 Gpx/s at 8 contexts against 5.0 in a process of its own (tools/archive/r4_stream_ctx_sweep.sh)
            frames.clear()
            for c in ctxs:
                c.close()
            ctxs.clear()
            e2e = end_to_end_leg(_lib, abi, host, synth, distinct[0], local_rank, npx)
        except Exception as e:
            e2e = {"error": repr(e)[:300]}

    # ---- roofline of the dominant kernel (restoration + colour stage) and of the whole path
    out_bytes_px = 6.0 if args.workload == "vardct8k_pq" else 12.0
    side = 21.0 / 64.0  # per-pixel share of the 8x8-cell side info
    path_bytes = (12.0 + out_bytes_px + side) * npx + 1.58e6  # whole path, algorithmic (24.5 B/px for f32 out)
    rest_bytes = (12.0 + out_bytes_px + 8.0 / 64.0) * npx     # restore stage: planes in + planes out + hfMul/sharpness
    idct_bytes = (12.0 + 12.0 + side) * npx + 1.58e6          # IDCT stage: coefficients in, planes out, side info, weights
    rest_s = ms_rest * 1e-3
    achieved = rest_bytes / rest_s / 1e9 if rest_s > 0 else 0.0
    path_gbs = path_bytes * fpg * args.steps / elapsed / 1e9  # per GPU
    epf_iters = real_stats["epf_iters"] if real_stats else args.epf_iters
    # measured per-launch counters of this kernel (profiles/<round>_traffic.json, written by tools/profile_round.sh from the
    # rocprofv3 --pmc passes of this same command): HBM bytes and VALU wave-instructions. rocprofv3 cannot run inside
    # the bench; the figures are carried over only when workload and variant match, else null.
    traffic, valu_insts, traffic_src, valu_all = None, None, None, None
    src_sha = kernel_source_sha()
    for name in ("r6_traffic.json", "r5_traffic.json", "r4_traffic.json", "r3_traffic.json", "r2_traffic.json", "r1_traffic.json"):
        tpath = os.path.join(ROOT, "profiles", name)
        if args.workload == "vardct4k" and epf_iters == 2 and args.mix == "default" and os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                # the counters belong to ONE build of the kernel: they are carried only while the kernel's sources still hash to
                # what was profiled (files without a stamp -- rounds 1 and 2 -- are stale by definition)
                if tj.get("kernel_source_sha256") != src_sha:
                    continue
                traffic = int(tj["hbm_bytes_per_launch"])
                valu_insts = tj.get("valu_wave_insts_per_launch")
                valu_all = tj.get("valu_wave_insts_all_launches_per_frame")
                traffic_src = "profiles/" + name
                break
            except Exception:
                pass
    roofline = {
        "bound": "valu", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_source": traffic_src,
        "kernel": "k_restore_fused (Gab, EPF x%d, XYB%s): the launch's own start / stop HIP events (hipExtLaunchKernel), frame 0 alone on the device (mean of 32 runs enqueued back to back)"
                  % (epf_iters, ", PQ + u16" if args.workload == "vardct8k_pq" else ""),
        "kernel_ms": round(ms_rest, 4), "restoration_stage_ms_events": round(ms_rest_stage, 4), "algorithmic_bytes_per_launch": int(rest_bytes),
        "kernel_ms_in_batch": round(ms_rest_b, 4),
        "idct_stage_ms": round(ms_idct, 4), "idct_stage_ms_in_batch": round(ms_idct_b, 4), "frame_ms_events": round(ms_all, 4),
        "idct_stage_GBps": round(idct_bytes / (ms_idct * 1e-3) / 1e9, 1) if ms_idct > 0 else None,
        "idct_stage_frac": round(idct_bytes / (ms_idct * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms_idct > 0 else None,
        "path_algorithmic_GBps": round(path_gbs, 1), "path_frac": round(path_gbs / HBM_PEAK_GBS, 4),
        "note": "achieved/peak/frac are the HBM figures the contract asks for (algorithmic bytes / launch time / 8 TB/s); the kernel's "
                "actual bound is VALU issue: the reference's summation order forbids FMA, so every multiply and add is its own "
                "instruction (valu_* fields: wave-instructions per launch from the PMC pass, against 1024 SIMDs x 1 per 2 cycles "
                "x 2.4 GHz)",
    }
    if valu_insts:
        g = valu_insts / rest_s / 1e9 if rest_s > 0 else 0.0
        roofline.update({"valu_wave_insts_per_launch": int(valu_insts), "valu_insts_per_pixel": round(valu_insts * 64.0 / npx, 1),
                         "valu_achieved_Ginst_s": round(g, 1), "valu_peak_Ginst_s": VALU_PEAK_GINST,
                         "valu_frac": round(g / VALU_PEAK_GINST, 4)})
    if valu_all:
        # the bound of the PATH as it is built today: the wave-instructions of ALL launches of a frame at the issue peak
        ceil_ms = valu_all / (VALU_PEAK_GINST * 1e9) * 1e3
        per_frame_ms = ms_per_step / max(fpg, 1)
        roofline.update({"valu_wave_insts_all_launches_per_frame": int(valu_all), "valu_ceiling_ms": round(ceil_ms, 4),
                         "valu_frac_path": round(ceil_ms / per_frame_ms, 4) if per_frame_ms > 0 else None,
                         "valu_frac_single_frame": round(ceil_ms / ms_all, 4) if ms_all > 0 else None,
                         "valu_note": "valu_ceiling_ms = wave-instructions of every launch of one frame / 1228.8 G per s (1024 SIMDs x 1 per 2 "
                                      "cycles x 2.4 GHz); valu_frac_path = that ceiling / the batch's time per frame"})

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
                         "C oracle = line-faithful restatement of the Java path" % (reps, W, H, ncores),
               "seconds": round(t_all * reps, 2),
               "value_1core": round(npx / t_1 / 1e6, 3), "seconds_1core": round(t_1, 2),
               "cpu_model": cpu_model(), "host_cores": os.cpu_count()}
        cpu.update(jvm_probe(os.path.join(ROOT, "tests", "golden", "samples", "lenna.jxl")))

    line = {
        "metric": "Mpixels/s decoded (VarDCT 4K frame) at 1/2/4/8 MI355X vs host-CPU jxlatte",
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic" if distinct else "real bitstream %s (parsed by the C++ front-end)" % os.path.basename(args.input),
        "config": {"workload": "%s: %d independent %dx%d VarDCT frames per GPU (mix=%s, Gab + EPF x%d + XYB, %s out), inputs resident in HBM"
                               % (args.workload, fpg, W, H, args.mix if distinct else "as coded", epf_iters,
                                  "f32" if out_bytes_px == 12.0 else "PQ u16"),
                   "frames_per_gpu": fpg, "frames_total": n_frames, "frame_seeds": "1000 + i, frame i on rank i mod %d" % world if n_frames > 1 else "1234",
                   "distinct_frames": len(distinct) or 1, "streams": 1 if args.streams == 1 else min(fpg, args.stream_groups) if args.stream_groups else fpg,
                   "varblock_area_share": synth.type_histogram(distinct[0]) if distinct else real_stats["varblocks"],
                   "kernel_launches_per_frame": launches,
                   "single_frame_ms": round(single_ms, 4),
                   "single_frame_Mpx_s": round(npx / single_ms / 1e3, 1), "input_gen_s": round(gen_s, 1)},
        "roofline": roofline,
        "cpu_baseline": cpu,
    }
    line["timing"] = timing
    if e2e:
        line["untimed"] = e2e
    if gather:
        line["gather"] = gather
    if (not args.no_also and world == 1 and not use_dist and args.workload == "vardct4k" and args.mix == "default" and args.epf_iters == 2
            and not args.size and args.stages == 31 and not args.batch and args.frames_per_gpu == 8 and args.streams == 0):
        # the contexts of the headline are released first: the side runs get the device to themselves
        for c in ctxs:
            c.close()
        line["also"] = also_runs(args)
    emit(line)
    if use_dist:
        dist.destroy_process_group()


def _normalised_source(path):
    """a source file without comments, blank lines and indentation: what the stamp of a profile hashes, so that a comment or a
    re-wrapped line does not void the counters of an unchanged kernel (r4: a switch added to a hashed file six minutes after the
    last profile pass cost the driver's line its traffic figure)"""
    import re
    txt = open(path, "r", encoding="utf-8", errors="replace").read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    out = []
    for ln in txt.split("\n"):
        ln = re.sub(r"//.*$", "", ln).strip()
        if ln:
            out.append(re.sub(r"\s+", " ", ln))
    return "\n".join(out).encode()


def kernel_source_sha(files=("k_restore_fused.hip", "restore_fused_body.h", "restore_sink.h", "jxl_fastpow.h")):
    """sha256 over the normalised sources of the dominant kernel (the stamp of profiles/rN_traffic.json)"""
    import hashlib
    h = hashlib.sha256()
    for f in files:
        h.update(_normalised_source(os.path.join(ROOT, "jxlatte_amd", "csrc", f)))
    return h.hexdigest()


MODULAR_SOURCES = ("k_modular.hip", "k_modular_vh.hip", "modular_tend.h")


def gather_legs(args, torch, dist, shard, lib, ctxs, frames, step, sync_all, n_frames, rank, world, H, W):
    """time (i) the gather of one step's output alone and (ii) K steps of compute with the gather of step k overlapping the
    compute of step k+1; both as max over ranks. The send tensor is [k, 3, H, W] of the output element type."""
    step()  # the output element size is known once a frame has run
    es = lib.jxl_vardct_out_elem_size(ctxs[0].h)
    k = len(ctxs)
    # f32 planes as float32 [k, 3, H, W]; narrower outputs as raw bytes (RCCL has no uint16): [k, 3, H, W * es] uint8
    bufs = [torch.empty((k, 3, H, W) if es == 4 else (k, 3, H, W * es), dtype=torch.float32 if es == 4 else torch.uint8, device="cuda")
            for _ in range(2)]
    frame_bytes = 3 * H * W * es
    cur = torch.cuda.current_stream()
    ext = [torch.cuda.ExternalStream(int(c.stream)) for c in ctxs]

    def stage_outputs(buf):
        # D2D copies on the ctx streams (asynchronous), then the torch stream (which RCCL orders itself after) waits for them
        for i, c in enumerate(ctxs):
            c.call("jxl_vardct_copy_output_device", C.c_void_p(buf.data_ptr() + i * frame_bytes))
            ev = torch.cuda.Event()
            ev.record(ext[i])
            cur.wait_event(ev)

    def maxtime(t):
        d = torch.tensor([t], dtype=torch.float64, device="cuda")
        dist.all_reduce(d, op=dist.ReduceOp.MAX)
        return float(d.item())

    step()
    stage_outputs(bufs[0])
    got = shard.gather_planes(bufs[0], n_frames, rank, world)  # warm-up: communicator, allocations
    sync_all()
    # what arrived on rank 0 must be what jxl_vardct_read_output hands back, frame by frame (checked on rank 0's own frames:
    # global indices rank, rank + world, ...)
    same = None
    if rank == 0 and got is not None:
        same = True
        for i, fr in enumerate(frames):
            ref = np.ascontiguousarray(fr.readOutput())
            mine = got[shard.frames_of_rank(n_frames, rank, world)[i]].cpu().numpy()
            same = same and np.array_equal(mine.reshape(-1).view(np.uint8), ref.reshape(-1).view(np.uint8))
    reps = 3
    g0 = time.perf_counter()
    for _ in range(reps):
        shard.gather_planes(bufs[0], n_frames, rank, world)
    torch.cuda.synchronize()
    alone = maxtime((time.perf_counter() - g0) / reps)
    sync_all()
    ksteps = max(3, min(args.steps, 10))
    g0 = time.perf_counter()
    for s in range(ksteps):
        step()
        buf = bufs[s & 1]
        # the ctx streams must not overwrite a send tensor RCCL may still be reading (gather of step s-2): the torch stream
        # has been ordered behind that gather, so make the ctx streams wait for the torch stream's position
        ev = torch.cuda.Event()
        ev.record(cur)
        for e in ext:
            e.wait_event(ev)
        stage_outputs(buf)
        shard.gather_planes(buf, n_frames, rank, world)
    torch.cuda.synchronize()
    overl = maxtime((time.perf_counter() - g0) / ksteps)
    npx_step = float(H) * W * n_frames
    return {"payload_MB_per_rank": round(k * frame_bytes / 1e6, 1), "out_elem_bytes": es, "gathered_equals_read_output": same,
            "gather_alone_ms": round(alone * 1e3, 3),
            "compute_plus_overlapped_gather_ms_per_step": round(overl * 1e3, 3),
            "compute_plus_overlapped_gather_Mpx_s": round(npx_step / overl / 1e6, 1),
            "note": "shard.gather_planes (ncclGather to rank 0 + reassembly in frame order); never inside the timed steps of `value`"}


def _group_call(planes, dtype):
    """ctypes arguments of one put_group call, built ahead of the timed loop"""
    ct = C.c_int16 if dtype == np.int16 else C.c_int32
    q = [np.ascontiguousarray(a, dtype) if a.dtype != dtype or not a.flags.c_contiguous else a for a in planes]
    pp = (C.POINTER(ct) * 3)(*[a.ctypes.data_as(C.POINTER(ct)) for a in q])
    strides = (C.c_int32 * 3)(*[a.shape[1] for a in q])
    return pp, strides, q


def end_to_end_leg(_lib, abi, host, synth, frame, device, npx):
    """One 4K frame through the boundary as the Java host would drive it, each part timed on the host clock (synchronous
    calls): set_lfgroup (host scatter) + jxl_vardct_prepare (varblock binning, CfL masks, side-table upload) = host_prepare;
    put_group of every group = H2D (pageable int32 planes, the reference's representation); run + read_output with the
    device output stage on (sRGB, RGB8 interleaved: 3 B/px back) = kernels + D2H."""
    c = _lib.Context(device)
    try:
        d = dict(frame)
        p = abi.VarDCTParams.from_buffer_copy(d["params"])
        p.transfer, p.out_format, p.stages = abi.TRANSFER_SRGB, abi.OUT_RGB8, 31
        best = None
        for rep in range(2):  # first repetition pays allocations; report the second
            t = {}
            a = time.perf_counter()
            fr = host.Frame(c, p, d["weights"], d["woffs"])
            t["begin_frame_ms"] = (time.perf_counter() - a) * 1e3
            a = time.perf_counter()
            for g in d["lfgroups"]:
                fr.setLFGroup(g)
            c.call("jxl_vardct_prepare")
            t["host_prepare_ms"] = (time.perf_counter() - a) * 1e3
            views = [synth.group_view(d, grp) for grp in range(synth.num_groups(d))]
            calls = [_group_call(v, np.int32) for v in views]  # argument marshalling is the binding's cost, not the path's
            a = time.perf_counter()
            for grp, (pp, strides, _) in enumerate(calls):
                c.call("jxl_vardct_put_group", 0, grp, pp, strides)
            c.synchronize()
            t["h2d_ms"] = (time.perf_counter() - a) * 1e3
            a = time.perf_counter()
            fr.run()
            c.synchronize()
            t["kernels_ms"] = (time.perf_counter() - a) * 1e3
            a = time.perf_counter()
            out = fr.readOutput()
            t["d2h_ms"] = (time.perf_counter() - a) * 1e3
            best = t
        tot = best["host_prepare_ms"] + best["h2d_ms"] + best["kernels_ms"] + best["d2h_ms"]
        res = {k: round(v, 3) for k, v in best.items()}
        res.update({"end_to_end_ms": round(tot, 3), "end_to_end_Mpx_s": round(npx / tot / 1e3, 1),
                    "h2d_MB": round(12.0 * npx / 1e6, 1), "d2h_MB": round(out.nbytes / 1e6, 1),
                    "note": "one 4K frame, host clock, synchronous C-ABI calls from pageable memory; PCIe-inclusive, never `value`"})
        # the same frame over the wire format built for this leg: int16 coefficients in page-locked buffers
        # (jxl_host_alloc + jxl_vardct_put_group_i16: direct DMA, no host wait per group), page-locked output
        lib = _lib.load()
        pins = []
        try:
            views = [synth.group_view(d, grp) for grp in range(synth.num_groups(d))]
            fits = all(int(np.abs(a).max()) < 32768 for v in views for a in v)
            pv = []
            for v in views:
                row = []
                for a in v:
                    pa = host.PinnedArray(lib, a.shape, np.int16 if fits else np.int32)
                    pa.array[...] = a
                    pins.append(pa)
                    row.append(pa.array)
                pv.append(row)
            pout = host.PinnedArray(lib, out.shape, out.dtype)
            pins.append(pout)
            t = {}
            for rep in range(2):
                fr = host.Frame(c, p, d["weights"], d["woffs"])
                for g in d["lfgroups"]:
                    fr.setLFGroup(g)
                c.call("jxl_vardct_prepare")
                calls = [_group_call(v, np.int16 if fits else np.int32) for v in pv]
                a = time.perf_counter()
                for grp, (pp, strides, _) in enumerate(calls):
                    c.call("jxl_vardct_put_group_i16" if fits else "jxl_vardct_put_group", 0, grp, pp, strides)
                c.synchronize()
                t["h2d_ms"] = (time.perf_counter() - a) * 1e3
                a = time.perf_counter()
                fr.run()
                c.synchronize()
                t["kernels_ms"] = (time.perf_counter() - a) * 1e3
                a = time.perf_counter()
                pp = (C.c_void_p * 3)(pout.array.ctypes.data, None, None)
                c.call("jxl_vardct_read_output", pp, fr.width)
                t["d2h_ms"] = (time.perf_counter() - a) * 1e3
            tot2 = best["host_prepare_ms"] + t["h2d_ms"] + t["kernels_ms"] + t["d2h_ms"]
            res["pinned_i16" if fits else "pinned_i32"] = dict(
                {k: round(v, 3) for k, v in t.items()}, h2d_MB=round((6.0 if fits else 12.0) * npx / 1e6, 1),
                end_to_end_ms=round(tot2, 3), end_to_end_Mpx_s=round(npx / tot2 / 1e3, 1),
                identical_output=bool(np.array_equal(pout.array, out)))
            # and with the groups written in place into the library's page-locked frame planes (jxl_vardct_map_coeffs_i16):
            # three DMA transfers per frame instead of three per group. Filling the planes stands for the entropy decoder's
            # own stores and is not part of the transfer time.
            if fits:
                t = {}
                for rep in range(2):
                    fr = host.Frame(c, p, d["weights"], d["woffs"])
                    for g in d["lfgroups"]:
                        fr.setLFGroup(g)
                    c.call("jxl_vardct_prepare")
                    a = time.perf_counter()
                    mp = fr.mapCoeffsI16(no_fill=True)  # r4: every group is written below, nothing to zero-fill
                    t["map_ms"] = (time.perf_counter() - a) * 1e3
                    for ch in range(3):
                        mp[ch][...] = d["coeff"][ch]
                    a = time.perf_counter()
                    fr.commitCoeffsI16(np.ones(synth_num_groups(d), np.uint8))
                    c.synchronize()
                    t["h2d_ms"] = (time.perf_counter() - a) * 1e3
                    a = time.perf_counter()
                    fr.run()
                    c.synchronize()
                    t["kernels_ms"] = (time.perf_counter() - a) * 1e3
                    a = time.perf_counter()
                    pp = (C.c_void_p * 3)(pout.array.ctypes.data, None, None)
                    c.call("jxl_vardct_read_output", pp, fr.width)
                    t["d2h_ms"] = (time.perf_counter() - a) * 1e3
                tot3 = best["host_prepare_ms"] + t["map_ms"] + t["h2d_ms"] + t["kernels_ms"] + t["d2h_ms"]
                res["mapped_i16"] = dict({k: round(v, 3) for k, v in t.items()}, h2d_MB=round(6.0 * npx / 1e6, 1),
                                         end_to_end_ms=round(tot3, 3), end_to_end_Mpx_s=round(npx / tot3 / 1e3, 1),
                                         identical_output=bool(np.array_equal(pout.array, out)))
        finally:
            for x in pins:
                x.free()
        # ---- streaming: frames in flight on several contexts, one host thread each, so that the host work of frame k+1
        #      (begin_frame, LF groups, prepare, zero-fill + the decoder's coefficient stores), the H2D of its planes, the kernels
        #      of frame k and the D2H of frame k-1 overlap. Every frame goes through the whole boundary; never `value`.
        try:
            if res.get("mapped_i16", {}).get("identical_output"):
                res["streaming"] = streaming_leg(_lib, host, d, p, device, npx, out)
        except Exception as e:
            res["streaming"] = {"error": repr(e)[:300]}
        return res
    finally:
        c.close()


def synth_num_groups(d):
    from jxlatte_amd import synth
    return synth.num_groups(d)


class _StreamBenchArgs(C.Structure):  # tools/native/stream_bench.cpp: jxl_stream_bench_args
    _fields_ = [("lib_path", C.c_char_p), ("device", C.c_int32), ("n_ctx", C.c_int32), ("frames_per_ctx", C.c_int32),
                ("params", C.c_void_p), ("weights", C.c_void_p), ("n_weights", C.c_size_t), ("woffs", C.c_void_p),
                ("lfgroups", C.c_void_p), ("n_lfgroups", C.c_int32), ("coeff", C.c_void_p * 3), ("rows", C.c_int32 * 3),
                ("cols", C.c_int32 * 3), ("group_written", C.c_void_p), ("n_groups", C.c_int32), ("outs", C.c_void_p),
                ("out_stride", C.c_int64), ("wall_s", C.c_double), ("phase_s", C.c_double * 8), ("err", C.c_char * 256)]


STREAM_PHASES = ("begin", "lfgroups", "prepare", "map", "stores", "commit", "wait_prev+run", "read_begin")


def streaming_leg_native(_lib, host, d, p, device, npx, ref_out, n_ctx, frames_per_ctx):
    """the streaming leg from native host threads (tools/native/stream_bench.cpp): the call sequence of streaming_leg() below,
    one std::thread per context -- what a JVM host's decoder threads do; Python threads serialise on the interpreter lock
    between the calls. None when the helper is not built."""
    from jxlatte_amd import abi
    so = os.path.join(ROOT, "jxlatte_amd", "libjxl_stream_bench.so")
    if not os.path.exists(so) or os.environ.get("JXL_BENCH_STREAM_PY"):
        return None
    drv = C.CDLL(so)
    drv.jxl_stream_bench.restype = C.c_int
    drv.jxl_stream_bench.argtypes = [C.POINTER(_StreamBenchArgs)]
    lib = _lib.load()
    coeff16 = [np.ascontiguousarray(a, np.int16) for a in d["coeff"]]
    weights = np.ascontiguousarray(d["weights"], np.float32)
    woffs = np.ascontiguousarray(d["woffs"], np.int32)
    descs = [abi.make_lfgroup_desc(g) for g in d["lfgroups"]]
    dptr = (C.c_void_p * len(descs))(*[C.addressof(x) for x in descs])
    written = np.ones(synth_num_groups(d), np.uint8)
    pouts = [host.PinnedArray(lib, ref_out.shape, ref_out.dtype) for _ in range(n_ctx)]
    try:
        optr = (C.c_void_p * n_ctx)(*[x.array.ctypes.data for x in pouts])
        a = _StreamBenchArgs()
        a.lib_path = _lib.SO_PATH.encode()
        a.device, a.n_ctx, a.frames_per_ctx = device, n_ctx, frames_per_ctx
        a.params = C.addressof(p)
        a.weights, a.n_weights, a.woffs = weights.ctypes.data, weights.size, woffs.ctypes.data
        a.lfgroups, a.n_lfgroups = C.addressof(dptr), len(descs)
        for ch in range(3):
            a.coeff[ch] = coeff16[ch].ctypes.data
            a.rows[ch], a.cols[ch] = coeff16[ch].shape
        a.group_written, a.n_groups = written.ctypes.data, written.size
        a.outs, a.out_stride = C.addressof(optr), p.width
        if drv.jxl_stream_bench(C.byref(a)) != 0:
            return {"error": a.err.decode("utf-8", "replace")}
        same = [bool(np.array_equal(x.array, ref_out)) for x in pouts]
    finally:
        for x in pouts:
            x.free()
    n = n_ctx * frames_per_ctx
    return {"contexts": n_ctx, "frames": n, "host_threads": "native (tools/native/stream_bench.cpp)", "wall_ms": round(a.wall_s * 1e3, 2),
            "ms_per_frame": round(a.wall_s * 1e3 / n, 3), "streaming_end_to_end_Mpx_s": round(npx * n / a.wall_s / 1e6, 1),
            "identical_output": all(same),
            "host_ms_per_frame_and_thread": dict(zip(STREAM_PHASES, [round(float(v) / n * 1e3, 2) for v in a.phase_s])),
            "note": "%d contexts, one native host thread each (a JVM host's decoder threads; no interpreter lock between the calls): "
                    "begin_frame + LF groups + prepare + map (no zero-fill: every group is written) + coefficient stores + commit + run + "
                    "read_output_begin per frame, read_output_wait one frame later (RGB8); all frames through the whole boundary; "
                    "PCIe-inclusive, never `value`" % n_ctx}


STREAM_HW_QUEUES = "16"  # GPU_MAX_HW_QUEUES of the streaming leg's process (the runtime's default is 4)


def streaming_leg(_lib, host, d, p, device, npx, ref_out, n_ctx=int(os.environ.get("JXL_BENCH_STREAM_CTX", "10")),
                  frames_per_ctx=int(os.environ.get("JXL_BENCH_STREAM_FRAMES", "24")), in_child=False):
    import threading
    if not in_child and not os.environ.get("JXL_BENCH_STREAM_INPROC"):
        # a process of its own: eight contexts and their streams share the runtime's hardware queues -- 4 by default, and a table
        # copy or an IDCT launch of one context then waits behind the other contexts' bus transfers (begin_frame 1.5-3.8 ms per
        # call; tools/archive/r5_stream_sections.sh). GPU_MAX_HW_QUEUES is read when the runtime initialises, so the leg that stands for
        # a multi-decoder host gets its own process with the setting INTEGRATION.md recommends; the timed step keeps the default
        # (it is 1-3 % slower with 16 queues).
        import pickle
        import tempfile
        with tempfile.TemporaryDirectory() as td:
            ref_path = os.path.join(td, "job.pkl")
            with open(ref_path, "wb") as f:
                pickle.dump({"frame": d, "params": bytes(p), "npx": npx, "ref": ref_out}, f, protocol=4)
            env = dict(os.environ, GPU_MAX_HW_QUEUES=os.environ.get("JXL_BENCH_STREAM_HW_QUEUES", STREAM_HW_QUEUES))
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--streaming-child", "%d,%d,%d,%s" % (device, n_ctx, frames_per_ctx, ref_path)],
                               env=env, capture_output=True, text=True, timeout=600)
        try:
            res = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            return {"error": "streaming child failed: " + (r.stderr or r.stdout)[-300:]}
        res["process"] = "own process, GPU_MAX_HW_QUEUES=%s" % env["GPU_MAX_HW_QUEUES"]
        return res
    nat = streaming_leg_native(_lib, host, d, p, device, npx, ref_out, n_ctx, frames_per_ctx)
    if nat is not None:
        if os.environ.get("JXL_BENCH_STREAM_BOTH"):  # diagnostics: the Python-thread form beside it
            os.environ["JXL_BENCH_STREAM_PY"] = "1"
            try:
                nat["python_threads"] = streaming_leg(_lib, host, d, p, device, npx, ref_out, n_ctx, frames_per_ctx)
            finally:
                del os.environ["JXL_BENCH_STREAM_PY"]
        return nat
    lib = _lib.load()
    coeff16 = [np.ascontiguousarray(a, np.int16) for a in d["coeff"]]
    all_groups = np.ones(synth_num_groups(d), np.uint8)
    ctxs = [_lib.Context(device) for _ in range(n_ctx)]
    pouts = [host.PinnedArray(lib, ref_out.shape, ref_out.dtype) for _ in range(n_ctx)]
    start = threading.Barrier(n_ctx + 1)
    errs, same, t_end = [], [True] * n_ctx, [0.0] * n_ctx
    phases = [np.zeros(8) for _ in range(n_ctx)]  # host time per call, summed per context
    t_beg, t_loop = [0.0] * n_ctx, [0.0] * n_ctx

    def worker(i):
        c = ctxs[i]
        try:
            pp = (C.c_void_p * 3)(pouts[i].array.ctypes.data, None, None)

            state = {"pending": False}

            def one_frame():
                # frame k+1's host share (begin ... commit) runs while frame k's kernels and output copy are in flight on the
                # same context: read_output is split into begin (queued) and wait (before the output buffer is reused)
                t = [time.perf_counter()]
                fr = host.Frame(c, p, d["weights"], d["woffs"]); t.append(time.perf_counter())
                for g in d["lfgroups"]:
                    fr.setLFGroup(g)
                t.append(time.perf_counter())
                c.call("jxl_vardct_prepare"); t.append(time.perf_counter())
                mp = fr.mapCoeffsI16(no_fill=True); t.append(time.perf_counter())
                for ch in range(3):
                    np.copyto(mp[ch], coeff16[ch])  # stands for the entropy decoder's stores (every group, zeros included)
                t.append(time.perf_counter())
                fr.commitCoeffsI16(all_groups); t.append(time.perf_counter())
                if state["pending"]:
                    c.call("jxl_vardct_read_output_wait")  # frame k's pixels have landed: the buffer is free again
                fr.run(); t.append(time.perf_counter())
                c.call("jxl_vardct_read_output_begin", pp, fr.width); t.append(time.perf_counter())
                state["pending"] = True
                phases[i] += np.diff(t)
            one_frame()  # allocations, page-locking
            c.call("jxl_vardct_read_output_wait")
            state["pending"] = False
            phases[i][:] = 0.0
            start.wait()
            t_beg[i] = time.perf_counter()
            for _ in range(frames_per_ctx):
                one_frame()
            t_loop[i] = time.perf_counter()
            c.call("jxl_vardct_read_output_wait")
            t_end[i] = time.perf_counter()
        except Exception as e:  # noqa: BLE001
            errs.append(repr(e)[:200])
            try:
                start.abort()
            except Exception:
                pass

    th = [threading.Thread(target=worker, args=(i,)) for i in range(n_ctx)]
    for t in th:
        t.start()
    try:
        start.wait()
        a = time.perf_counter()
        for t in th:
            t.join()
        wall = (max(t_end) if min(t_end) > 0.0 else time.perf_counter()) - a
        # the bench's own check, not the boundary's work -- and made only when every thread has finished: a 25 MB comparison in
        # the thread that finishes first costs the threads still running their share of the host's memory system (r4)
        same = [bool(np.array_equal(x.array, ref_out)) for x in pouts]
        if os.environ.get("JXL_BENCH_STREAM_DEBUG"):
            print("stream debug: a->beg %s  beg->loop %s  loop->end %s" % (["%.1f" % ((x - a) * 1e3) for x in t_beg], ["%.1f" % ((y - x) * 1e3) for x, y in zip(t_beg, t_loop)],
                                                                          ["%.1f" % ((y - x) * 1e3) for x, y in zip(t_loop, t_end)]), file=sys.stderr)
    finally:
        for t in th:
            t.join()
        for x in pouts:
            x.free()
        for c in ctxs:
            c.close()
    if errs:
        return {"error": errs[0]}
    n = n_ctx * frames_per_ctx
    return {"contexts": n_ctx, "frames": n, "wall_ms": round(wall * 1e3, 2), "ms_per_frame": round(wall * 1e3 / n, 3),
            "streaming_end_to_end_Mpx_s": round(npx * n / wall / 1e6, 1), "identical_output": all(same),
            "host_threads": "python",
            "host_ms_per_frame_and_thread": dict(zip(STREAM_PHASES, [round(float(v), 2) for v in sum(phases) / n * 1e3])),
            "note": "%d contexts, one host thread each: begin_frame + LF groups + prepare + map (no zero-fill: every group is written) + "
                    "coefficient stores + commit (3 DMA transfers of int16 planes) + run + read_output_begin per frame, read_output_wait "
                    "one frame later (RGB8); all frames through the whole boundary; PCIe-inclusive, never `value`" % n_ctx}


def bench_modular(args, rank, world, local_rank, torch, dist):
    from jxlatte_amd import _lib, host, synth
    W, H = (1920, 1080) if args.workload == "modular1080p" else (7680, 4320)
    fpg = max(1, min(args.frames_per_gpu, int(os.environ.get("JXL_BENCH_MODULAR_MAX", "4"))))
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
    # HBM bytes of one plan from the PMC passes of tools/profile_modular.sh (carried while the kernels' sources hash to what was profiled)
    mtraffic, mtraffic_src = None, None
    for name in ("r6_modular_traffic.json", "r5_modular_traffic.json"):
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", name)))
            if tj.get("kernel_source_sha256") == kernel_source_sha(MODULAR_SOURCES) and args.workload in tj.get("plans", {}):
                mtraffic = int(tj["plans"][args.workload]["hbm_bytes_per_plan"])
                mtraffic_src = "profiles/" + name
                break
        except Exception:
            pass
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
               "sample": "%d x 1 image %dx%dx3 (mean), C oracle (H steps OpenMP over rows)" % (reps, W, H), "seconds": round(t * reps, 3),
               "cpu_model": cpu_model()}
    emit({
        "metric": "Mpixels/s inverse Squeeze (Modular %dx%d, 3 channels, default squeeze plan)" % (W, H),
        "value": round(value, 1), "unit": "Mpixels/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(elapsed * 1e3 / args.steps, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "int32", "data": "synthetic",
        "config": {"workload": "%s: %d images per GPU, %d squeeze steps" % (args.workload, fpg, len(mod["sp"])),
                   "launches": ctxs[0].lib.jxl_modular_last_launch_count(ctxs[0].h)},
        "roofline": {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                     "frac": round(gbs / HBM_PEAK_GBS, 4), "traffic": mtraffic, "traffic_source": mtraffic_src,
                     "algorithmic_bytes_per_image": int(bytes_img),
                     "note": "whole step list (24 B/px algorithmic over the elapsed time of all steps); V + H step of a level as one launch "
                             "(V output in LDS only), guessed chain states verified and the plan redone in order where one differs (DESIGN.md 4.3)"},
        "cpu_baseline": cpu,
    })


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""profiles/rN_rocprof_summary.md, rN_kernel_stats.csv, rN_traffic.json from the CSVs of tools/profile_round.sh:
    python tools/make_profile_summary.py r1 gpurun_out/prof_r1"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

tag, src = sys.argv[1], sys.argv[2]
out_dir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def find(sub, pat):
    f = glob.glob(os.path.join(src, sub, "**", pat), recursive=True)
    return f[0] if f else None


def short(n):
    n = n.replace("void ", "").replace("jxl::(anonymous namespace)::", "").replace("(anonymous namespace)::", "").replace("jxl::", "")
    return n.split("(")[0]


stats = find("stats", "*kernel_stats.csv")
rows = [r for r in csv.DictReader(open(stats)) if "rocclr" not in r["Name"]]
shutil.copy(stats, os.path.join(out_dir, "%s_kernel_stats.csv" % tag))


def counters(sub):
    f = find(sub, "*counter_collection.csv")
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    if f:
        for r in csv.DictReader(open(f)):
            acc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {n: sum(v) / len(v) for n, v in d.items()} for k, d in acc.items()}


def durations(sub):
    f = find(sub, "*kernel_trace.csv")
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[short(r["Kernel_Name"])].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    return {k: sum(v) / len(v) for k, v in acc.items()}


fetch, write, sq, lds = counters("fetch"), counters("write"), counters("sq"), counters("lds")
dur = durations("sq")
L = ["# %s -- rocprofv3 summaries (MI355X, gfx950, ROCm 7.2)" % tag, "",
     "Produced by `tools/profile_round.sh %s` on the GPU box + `tools/make_profile_summary.py` (commands inside the script;" % tag,
     "PMC counters in separate runs with --kernel-trace only).", "",
     "## 1. kernel-trace --stats of the default bench (8 frames/step on 2 shared streams -- two frames in flight: kernels of different frames overlap, so these",
     "   averages are LONGER than a kernel alone on the device; bench.py's `roofline.kernel_ms` is the isolated figure)", "",
     "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
for r in rows:
    L.append("| %s | %s | %.1f | %.2f | %s |" % (short(r["Name"]), r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
L += ["", "## 2. one frame alone on the device (frames-per-gpu 1): per-launch duration and PMC counters (mean over dispatches)", "",
      "FETCH_SIZE / WRITE_SIZE in KiB. WRITE_SIZE is exact for 16-byte-per-lane stores; FETCH_SIZE under-reports wide coalesced",
      "reads by 2x on gfx950 (the guide) and is uncalibrated for 4-byte-per-lane loads (the restoration kernel's tile load).", "",
      "| kernel | launch us (profiled) | FETCH KiB | WRITE KiB | waves | VALU insts | VALU busy quad-cycles | wave quad-cycles | wait_any | SALU insts | LDS conflict / active |",
      "|---|---|---|---|---|---|---|---|---|---|---|"]
for k in sorted(dur, key=lambda k: -dur[k]):
    if "rocclr" in k:
        continue
    s, l = sq.get(k, {}), lds.get(k, {})
    L.append("| %s | %.1f | %.0f | %.0f | %.0f | %.3g | %.3g | %.3g | %.3g | %.3g | %.3g / %.3g |" % (
        k, dur[k], fetch.get(k, {}).get("FETCH_SIZE", 0), write.get(k, {}).get("WRITE_SIZE", 0), s.get("SQ_WAVES", 0), s.get("SQ_INSTS_VALU", 0),
        s.get("SQ_ACTIVE_INST_VALU", 0), s.get("SQ_WAVE_CYCLES", 0), s.get("SQ_WAIT_ANY", 0), l.get("SQ_INSTS_SALU", 0),
        l.get("SQ_LDS_BANK_CONFLICT", 0), l.get("SQ_LDS_IDX_ACTIVE", 0)))
L += ["", "IDCT stage kernels: SQ_WAIT_ANY share of the wave cycles (waves parked in s_waitcnt / barriers):", ""]
for k in sorted(dur, key=lambda k: -dur[k]):
    if k.startswith(("k_idct", "k_llf")) and k in sq:
        L.append("* `%s`: %.0f %% (%.1f us per launch)" % (k, 100 * sq[k].get("SQ_WAIT_ANY", 0) / max(sq[k].get("SQ_WAVE_CYCLES", 1), 1), dur[k]))
# read over-fetch of the IDCT launches: FETCH_SIZE (x2: gfx950 counts a 128-byte request as 64) against the coefficient bytes the
# launch's varblock types own (12 bytes per pixel of their area share; the share comes from the bench line of the same round)
share_path = os.path.join(out_dir, "%s_bench_vardct4k.json" % tag)
if os.path.exists(share_path):
    try:
        cfg = json.loads(open(share_path).read().strip().splitlines()[-1])["config"]["varblock_area_share"]
        cls = {"k_idct_wg3<false>": ("DCT8", "DCT16", "DCT32", "DCT16_8", "DCT8_16", "DCT32_8", "DCT8_32", "DCT32_16", "DCT16_32"),
               "k_idct_wg3<true>": ("DCT64", "DCT64_32", "DCT32_64"),
               "k_idct_special_wg": ("HORNUSS", "DCT2", "DCT4", "DCT4_8", "DCT8_4", "AFV0", "AFV1", "AFV2", "AFV3")}
        L += ["", "Coefficient reads of the IDCT launches (4K frame = 99.5 MB of int32 coefficients; FETCH_SIZE doubled as the guide prescribes):", "",
              "| launch | area share | coefficient bytes used (MB) | fetched (MB) | ratio |", "|---|---|---|---|---|"]
        for k, types in cls.items():
            if k not in fetch:
                continue
            sh = sum(cfg.get(t, 0.0) for t in types)
            used = sh * 12.0 * 3840 * 2160 / 1e6
            got = 2 * fetch[k]["FETCH_SIZE"] * 1024 / 1e6
            L.append("| `%s` | %.3f | %.1f | %.1f | %.2f |" % (k, sh, used, got, got / used if used else 0))
        L += ["", "(the fetched figure includes the weight rows, block records and LF samples of the items: 1-3 MB per launch)"]
    except Exception as e:  # noqa: BLE001
        L += ["", "(over-fetch table not produced: %r)" % (e,)]
rk = sorted([k for k in dur if k.startswith("k_restore_fused")], key=lambda k: (", 0, 1>" not in k, -dur[k]))[0]  # the float-plane variant
fk, wk = fetch[rk]["FETCH_SIZE"], write[rk]["WRITE_SIZE"]
traffic = {"kernel": rk, "fetch_KiB": fk, "write_KiB": wk, "launch_us_profiled": dur[rk], "fetch_bytes_raw": fk * 1024, "write_bytes": wk * 1024,
           "hbm_bytes_per_launch": 2 * fk * 1024 + wk * 1024,
           "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (MI355X_MICROARCH.md, HBM): FETCH_SIZE doubled "
                     "(gfx950 counts 128-B requests as 64 B), WRITE_SIZE as read; per launch, 4K frame, Gab+EPFx2+XYB"}
s = sq[rk]
# the stamp bench.py checks before it carries these counters: sha256 over the kernel's sources as they are NOW -- run this script
# from the same tree the profiled library was built from
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
traffic["kernel_source_sha256"] = bench.kernel_source_sha()
traffic["valu_wave_insts_per_launch"] = s["SQ_INSTS_VALU"]
# every launch of ONE frame of the TIMED path: the IDCT-stage kernels and the restoration kernel, each launched once per frame (the
# boundary legs of the same process -- widening, put_group, the RGB8 sink -- are not part of a step): mean SQ_INSTS_VALU per dispatch
per_frame = {k: v["SQ_INSTS_VALU"] for k, v in sq.items() if "SQ_INSTS_VALU" in v and (k == rk or k.startswith(("k_idct", "k_llf", "k_large")))}
traffic["valu_wave_insts_all_launches_per_frame"] = sum(per_frame.values())
traffic["valu_wave_insts_by_kernel_per_frame"] = {k: round(v) for k, v in sorted(per_frame.items(), key=lambda kv: -kv[1])}
traffic["wait_any_share_of_wave_cycles"] = s.get("SQ_WAIT_ANY", 0) / max(s.get("SQ_WAVE_CYCLES", 1), 1)
json.dump(traffic, open(os.path.join(out_dir, "%s_traffic.json" % tag), "w"), indent=1)
lds_s = lds.get(rk, {})
ISSUE_CYCLES = 2.6  # cycles one wave64 f32 VALU instruction occupies its SIMD with 8 waves per SIMD: tools/ubench/pk_rate.hip
valu_us = s["SQ_INSTS_VALU"] * ISSUE_CYCLES / 1024 / 2.4e3
L += ["", "## 3. the dominant kernel", "",
      "`%s`: %.1f us per launch under the profiler; VALU instructions %.3g per 4K frame = %.0f per output pixel."
      % (rk, dur[rk], s["SQ_INSTS_VALU"], s["SQ_INSTS_VALU"] * 64 / (3840 * 2160)),
      "At the measured peak issue rate (%.1f cycles per wave-instruction and SIMD, tools/ubench/pk_rate.hip; packed f32 issues at half that rate,"
      % ISSUE_CYCLES,
      "so it buys this kernel nothing) that is %.1f us of VALU time over 1024 SIMDs at 2.4 GHz = %.0f %% of the launch. LDS: %.3g active cycles (%.1f us per CU),"
      % (valu_us, 100 * valu_us / dur[rk], lds_s.get("SQ_LDS_IDX_ACTIVE", 0), lds_s.get("SQ_LDS_IDX_ACTIVE", 0) / 256 / 2.4e3),
      "%.3g of them bank conflicts. HBM bytes per launch (corrected): %.1f MB vs %.1f MB algorithmic (%.1f us at 8 TB/s)."
      % (lds_s.get("SQ_LDS_BANK_CONFLICT", 0), traffic["hbm_bytes_per_launch"] / 1e6, (3840 * 2160 * 24 + 129600 * 8) / 1e6,
         (3840 * 2160 * 24 + 129600 * 8) / 8e6)]
open(os.path.join(out_dir, "%s_rocprof_summary.md" % tag), "w").write("\n".join(L) + "\n")
print("\n".join(L[-4:]))

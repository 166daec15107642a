#!/bin/bash
# rocprofv3 passes of the Modular bench lines (GPU box): tools/profile_modular.sh r5
#   --kernel-trace --stats per workload, then FETCH_SIZE and WRITE_SIZE in passes of their own (one image per step)
# -> gpurun_out/<R>_modular_kernel_stats.md and gpurun_out/<R>_modular_traffic.json (copy both into profiles/)
R=${1:-r5}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_mod_$R
cd /tmp && export TMPDIR=/tmp
for w in modular1080p modular8k; do
  timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -o p -- python3 $ROOT/bench.py --workload $w --frames-per-gpu 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/$w/fetch -o p -- python3 $ROOT/bench.py --workload $w --frames-per-gpu 1 --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  timeout 200 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/$w/write -o p -- python3 $ROOT/bench.py --workload $w --frames-per-gpu 1 --steps 4 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
done
cd $ROOT
R=$R OUT=$OUT python3 - <<'PY'
import csv, glob, json, os, sys, collections
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import bench
R, OUT, ROOT = os.environ["R"], os.environ["OUT"], os.environ["GRAFT_REPO_ROOT"]
out = ["# %s -- rocprofv3 of the Modular bench lines (1 image per step)" % R, ""]
plans = {}
for w, npx in (("modular1080p", 1920 * 1080), ("modular8k", 7680 * 4320)):
    f = glob.glob("%s/%s/**/*kernel_stats.csv" % (OUT, w), recursive=True)
    out += ["## bench.py --workload %s --frames-per-gpu 1" % w, "", "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    calls = {}
    for r in csv.DictReader(open(f[0])) if f else []:
        if "rocclr" in r["Name"]: continue
        n = r["Name"].replace("void ", "").replace("jxl::", "").split("(")[0]
        calls[n] = int(r["Calls"])
        out.append("| %s | %s | %.1f | %.2f | %s |" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    out.append("")
    # HBM bytes of one plan: the counters of every launch of the pass, divided by the plans the pass ran (5: 1 warm-up + 4 steps)
    tot = {}
    for sub, cn in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
        ff = glob.glob("%s/%s/%s/**/*counter_collection.csv" % (OUT, w, sub), recursive=True)
        acc, by = 0.0, collections.defaultdict(float)
        for r in csv.DictReader(open(ff[0])) if ff else []:
            if r["Counter_Name"] != cn or "rocclr" in r["Kernel_Name"]: continue
            acc += float(r["Counter_Value"]); by[r["Kernel_Name"].replace("void ", "").replace("jxl::", "").split("(")[0]] += float(r["Counter_Value"])
        tot[cn] = acc / 5 * 1024  # KiB -> bytes, per plan
        tot[cn + "_by_kernel_MB_per_plan"] = {k: round(v / 5 * 1024 / 1e6, 2) for k, v in sorted(by.items(), key=lambda kv: -kv[1])}
    if tot.get("FETCH_SIZE") is not None and tot.get("WRITE_SIZE") is not None:
        hbm = 2 * tot["FETCH_SIZE"] + tot["WRITE_SIZE"]  # FETCH_SIZE doubled: gfx950 tallies 128-byte requests at 64 (MI355X_MICROARCH.md, HBM)
        alg = 24.0 * npx
        plans[w] = {"hbm_bytes_per_plan": hbm, "fetch_bytes_raw": tot["FETCH_SIZE"], "write_bytes": tot["WRITE_SIZE"], "algorithmic_bytes": alg,
                    "ratio": hbm / alg, "fetch_by_kernel_raw_MB": tot["FETCH_SIZE_by_kernel_MB_per_plan"], "write_by_kernel_MB": tot["WRITE_SIZE_by_kernel_MB_per_plan"]}
        out += ["HBM bytes per plan (FETCH_SIZE x 2 + WRITE_SIZE, separate passes): **%.1f MB** = %.2f x the algorithmic %.1f MB (read every input sample once, write every output sample once)."
                % (hbm / 1e6, hbm / alg, alg / 1e6), ""]
json.dump({"kernel_source_sha256": bench.kernel_source_sha(bench.MODULAR_SOURCES), "plans": plans,
           "method": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes over bench.py --workload <w> --frames-per-gpu 1 --steps 4 --warmup 1; "
                     "all launches of the pass summed and divided by 5 plans; FETCH_SIZE doubled (gfx950)"},
          open("%s/gpurun_out/%s_modular_traffic.json" % (ROOT, R), "w"), indent=1)
open("%s/gpurun_out/%s_modular_kernel_stats.md" % (ROOT, R), "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY

#!/bin/bash
# rocprofv3 --kernel-trace --stats of the modular bench lines (GPU box): tools/profile_modular.sh r1
R=${1:-r1}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_mod_$R
cd /tmp && export TMPDIR=/tmp
for w in modular1080p modular8k; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/$w -o p -- python3 $ROOT/bench.py --workload $w --frames-per-gpu 1 --no-cpu-baseline > /dev/null 2>&1
done
python3 - <<PY
import csv, glob, os
out = ["# $R -- rocprofv3 --kernel-trace --stats of the Modular bench lines (1 image per step, 23 runs each)", ""]
for w in ("modular1080p", "modular8k"):
    f = glob.glob("$OUT/%s/**/*kernel_stats.csv" % w, recursive=True)[0]
    out += ["## bench.py --workload %s --frames-per-gpu 1" % w, "", "| kernel | calls | avg us | total ms | % |", "|---|---|---|---|---|"]
    for r in csv.DictReader(open(f)):
        if "rocclr" in r["Name"]: continue
        n = r["Name"].replace("void ", "").replace("jxl::", "").split("(")[0]
        out.append("| %s | %s | %.1f | %.2f | %s |" % (n, r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6, r["Percentage"]))
    out.append("")
open("$ROOT/gpurun_out/${R}_modular_kernel_stats.md", "w").write("\n".join(out) + "\n")
print("\n".join(out))
PY

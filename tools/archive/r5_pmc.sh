#!/bin/bash
# generic counter passes over the 8K Modular bench line; every argument is one pass (a quoted counter list), each under its own timeout:
#   tools/r5_pmc.sh "TA_TA_BUSY_sum TA_BUFFER_TOTAL_CYCLES_sum" "TCC_REQ_sum TCC_HIT_sum"        (env: WL=modular8k, ENVV="A=1 B=2")
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${WL:-modular8k}
for e in $ENVV; do export $e; done
i=0
for pass in "$@"; do
  i=$((i+1)); rm -rf $R/gpurun_out/pmc_g/$i
  timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc $pass -d $R/gpurun_out/pmc_g/$i -o p -- python3 $R/bench.py --workload $WL --frames-per-gpu 1 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1 || echo "pass $i ($pass) failed or timed out"
  I=$i python3 - <<'PY'
import csv, os, collections
R=os.environ["GRAFT_REPO_ROOT"]; i=os.environ["I"]
try: rows=list(csv.DictReader(open(R+"/gpurun_out/pmc_g/%s/p_counter_collection.csv"%i)))
except Exception as e: print("no counters", e); raise SystemExit
by=collections.OrderedDict()
for r in rows:
    if not any(k in r["Kernel_Name"] for k in ("squeeze", "k_inv_vh")): continue
    by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"].replace("jxl::","").replace("void ","")[:18], r.get("Grid_Size","")), {})[r["Counter_Name"]]=float(r["Counter_Value"])
for (d,k,g),c in sorted(by.items())[-4:]:
    print(k, "grid", g, " ".join("%s=%.4g"%kv for kv in sorted(c.items())))
PY
done

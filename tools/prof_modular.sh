cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/prof_mod
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_mod -o p -- python3 $R/bench.py --workload modular8k --frames-per-gpu 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
rows=list(csv.DictReader(open(R+"/gpurun_out/prof_mod/p_kernel_trace.csv")))
rows=[r for r in rows if "squeeze" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
# last image: last 22 launches
last=rows[-36:]
t0=int(last[0]["Start_Timestamp"])
for r in last:
    print("%-28s grid %6s wg %4s  start %8.1f us  dur %7.1f us" % (r["Kernel_Name"][:28], r["Grid_Size_X"] if "Grid_Size_X" in r else r.get("Grid_Size",""), r.get("Workgroup_Size_X", r.get("Workgroup_Size","")), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
print("total span %.1f us" % ((int(last[-1]["End_Timestamp"])-t0)/1e3))
PY

# per-launch timeline of the last image of a Modular plan: tools/prof_modular.sh [modular8k|modular1080p] [n_last] [ENV=..]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
WL=${1:-modular8k}
NL=${2:-36}
[ -n "$3" ] && export $3
rm -rf $R/gpurun_out/prof_mod
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/prof_mod -o p -- python3 $R/bench.py --workload $WL --frames-per-gpu 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
NL=$NL python3 - <<'PY'
import csv, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
rows=list(csv.DictReader(open(R+"/gpurun_out/prof_mod/p_kernel_trace.csv")))
rows=[r for r in rows if "squeeze" in r["Kernel_Name"] or "k_inv_vh" in r["Kernel_Name"] or "modular" in r["Kernel_Name"] or "k_rct" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
n=len(rows)//4  # 1 warm-up + 3 steps
last=rows[-n:]
t0=int(last[0]["Start_Timestamp"])
for r in last:
    print("%-40s grid %8s,%3s,%3s wg %4s vgpr %3s lds %6s start %8.1f us  dur %7.1f us" % (r["Kernel_Name"].replace("jxl::","").replace("void ","")[:40], r.get("Grid_Size_X",""), r.get("Grid_Size_Y",""), r.get("Grid_Size_Z",""), r.get("Workgroup_Size_X", ""), r.get("VGPR_Count","?"), r.get("LDS_Block_Size", "?"), (int(r["Start_Timestamp"])-t0)/1e3, (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3))
print("launches %d total span %.1f us" % (len(last), (int(last[-1]["End_Timestamp"])-t0)/1e3))
PY

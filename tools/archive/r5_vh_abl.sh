#!/bin/bash
# timing-only ablations of the fused squeeze kernel (experiment build libjxlatte_amd_abl.so: python -m jxlatte_amd.build --tag=abl -DJXL_VH_ABL)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT
export JXL_AMD_LIB=$ROOT/jxlatte_amd/libjxlatte_amd_abl.so
cd /tmp; export TMPDIR=/tmp
for a in 0 1 2 3; do
  rm -rf $ROOT/gpurun_out/prof_abl
  JXL_VH_ABL=$a timeout 120 rocprofv3 --kernel-trace --output-format csv -d $ROOT/gpurun_out/prof_abl -o p -- python3 $ROOT/bench.py --workload modular8k --frames-per-gpu 1 --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  A=$a python3 - <<'PY'
import csv, os
R=os.environ["GRAFT_REPO_ROOT"]
rows=[r for r in csv.DictReader(open(R+"/gpurun_out/prof_abl/p_kernel_trace.csv")) if "k_inv_vh" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
last=rows[-3:]
print("abl=%s (1 no loads, 2 no stores): last three fused launches " % os.environ["A"] + "  ".join("%.1f us" % ((int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e3) for r in last))
PY
done

cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_mod
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/pmc_mod/a -o p -- python3 $R/bench.py --workload modular8k --frames-per-gpu 1 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA -d $R/gpurun_out/pmc_mod/b -o p -- python3 $R/bench.py --workload modular8k --frames-per-gpu 1 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for sub in "ab":
    rows=list(csv.DictReader(open(R+"/gpurun_out/pmc_mod/%s/p_counter_collection.csv"%sub)))
    # the last dispatches of each kernel = the full-size steps
    by=collections.OrderedDict()
    for r in rows:
        if "squeeze" not in r["Kernel_Name"]: continue
        by.setdefault((r["Dispatch_Id"], r["Kernel_Name"][:22], r.get("Grid_Size","")), {})[r["Counter_Name"]]=float(r["Counter_Value"])
    items=list(by.items())[-4:]
    for (d,k,g),c in items:
        print(k, "grid", g, " ".join("%s=%.3g"%kv for kv in sorted(c.items())))
PY

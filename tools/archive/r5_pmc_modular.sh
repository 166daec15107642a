# PMC counters of the big launches of the 8K Modular plan: tools/r5_pmc_modular.sh [ENV=..]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
[ -n "$1" ] && export $1
rm -rf $R/gpurun_out/pmc_mod
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_BUSY_CYCLES -d $R/gpurun_out/pmc_mod/a -o p -- python3 $R/bench.py --workload modular8k --frames-per-gpu 1 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $R/gpurun_out/pmc_mod/b -o p -- python3 $R/bench.py --workload modular8k --frames-per-gpu 1 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
timeout 120 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/pmc_mod/c -o p -- python3 $R/bench.py --workload modular8k --frames-per-gpu 1 --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
python3 - <<'PY'
import csv, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
for sub in "abc":
    try: rows=list(csv.DictReader(open(R+"/gpurun_out/pmc_mod/%s/p_counter_collection.csv"%sub)))
    except Exception as e: print(sub, "failed", e); continue
    by=collections.OrderedDict()
    for r in rows:
        if not any(k in r["Kernel_Name"] for k in ("squeeze", "k_inv_vh")): continue
        by.setdefault((int(r["Dispatch_Id"]), r["Kernel_Name"].replace("jxl::","").replace("void ","")[:22], r.get("Grid_Size","")), {})[r["Counter_Name"]]=float(r["Counter_Value"])
    items=sorted(by.items())[-4:]
    for (d,k,g),c in items:
        print(k, "grid", g, " ".join("%s=%.4g"%kv for kv in sorted(c.items())))
PY

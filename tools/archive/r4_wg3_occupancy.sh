#!/bin/bash
# where the waves of k_idct_wg3<false> wait at 2 and at 4 workgroups per CU (GPU box): tools/r4_wg3_occupancy.sh [type]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4/occ
mkdir -p $OUT
T=${1:-DCT8}
cd /tmp && export TMPDIR=/tmp
for G in 512 1024; do
  export JXL_WG3_GRID=$G
  rm -rf $OUT/$G
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT/$G/a -o p -- python3 $ROOT/tools/idct_types.py --types $T --mixes "" --reps 4 > $OUT/$G.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_IFETCH -d $OUT/$G/b -o p -- python3 $ROOT/tools/idct_types.py --types $T --mixes "" --reps 4 > $OUT/$G.b.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_VALU SQ_INSTS_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_THREAD_CYCLES_VALU SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_LEVEL_WAVES -d $OUT/$G/c -o p -- python3 $ROOT/tools/idct_types.py --types $T --mixes "" --reps 4 > $OUT/$G.c.log 2>&1
  python3 - <<PY
import csv,collections,glob
acc=collections.defaultdict(list); dur=[]
for f in glob.glob("$OUT/$G/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_idct_wg3<false>" in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for f in glob.glob("$OUT/$G/a/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "k_idct_wg3<false>" in r['Kernel_Name']:
            dur.append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
m={c:sum(x)/len(x) for c,x in acc.items()}
print("$T grid $G: launch %.1f us" % (sum(dur)/max(1,len(dur))), {c: float("%.4g" % v) for c,v in sorted(m.items())})
PY
done

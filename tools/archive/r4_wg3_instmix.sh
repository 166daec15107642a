#!/bin/bash
# dynamic instruction mix of the IDCT launches for frames of one type (GPU box): tools/r4_wg3_instmix.sh [types]
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4/instmix
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for T in ${1:-DCT8 DCT16 DCT32}; do
  rm -rf $OUT/$T
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_WAVE_CYCLES -d $OUT/$T/a -o p -- python3 $ROOT/tools/idct_types.py --types $T --mixes "" --reps 4 > $OUT/$T.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_BRANCH SQ_INSTS_SENDMSG SQ_INSTS_FLAT SQ_INSTS_GDS SQ_INSTS_EXP_GDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_ANY SQ_INSTS_WAVE32_LDS -d $OUT/$T/b -o p -- python3 $ROOT/tools/idct_types.py --types $T --mixes "" --reps 4 > $OUT/$T.b.log 2>&1
  python3 - <<PY
import csv,collections,glob
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/$T/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "idct" in r['Kernel_Name']:
            acc[r['Kernel_Name'][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
for k,v in acc.items():
    m={c:sum(x)/len(x) for c,x in v.items()}
    w=m.get('SQ_WAVES',1)
    print("$T", k, "waves %d" % w, " per wave:", {c: round(x/w,1) for c,x in m.items() if c!='SQ_WAVES'})
PY
done

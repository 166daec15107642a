#!/bin/bash
# SQ_INSTS_VALU / SQ_WAVE_CYCLES of one frame's kernels under two environments: tools/r3_valu_ab.sh "ENV=.." "-"
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3v
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for e in "$@"; do
  i=$((i+1))
  if [ "$e" != "-" ]; then export $e; fi
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU -d $OUT/v_$i -o p -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --frames-per-gpu 1 --no-end-to-end --no-gather > $OUT/v_$i.log 2>&1
  if [ "$e" != "-" ]; then unset ${e%%=*}; fi
  python3 - <<PY
import csv,collections
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open("$OUT/v_$i/p_counter_collection.csv")):
    acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
dur=collections.defaultdict(list)
for r in csv.DictReader(open("$OUT/v_$i/p_kernel_trace.csv")):
    dur[r['Kernel_Name'][:60]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
print("== [$e]")
for k,v in acc.items():
    if 'restore' in k or 'idct' in k:
        d=sorted(dur[k]); print("  %-62s us %.1f  %s" % (k, d[len(d)//2], {c:'%.4g'%(sum(x)/len(x)) for c,x in v.items()}))
PY
done

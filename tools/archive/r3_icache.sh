#!/bin/bash
# instruction-cache counters of the IDCT launches: default mix against single-type frames (one frame alone on the device)
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r3i
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $OUT/counters.txt 2>&1
for MIX in default dct8; do
  ONE="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --frames-per-gpu 1 --mix $MIX"
  rocprofv3 --kernel-trace --output-format csv --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH -d $OUT/ic_$MIX -o p -- $ONE > $OUT/ic_$MIX.log 2>&1
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVE_CYCLES SQ_INST_CYCLES_VMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SMEM -d $OUT/w_$MIX -o p -- $ONE > $OUT/w_$MIX.log 2>&1
done
find $OUT -name "*counter_collection.csv" | head

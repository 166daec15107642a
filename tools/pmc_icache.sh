#!/bin/bash
# instruction-cache counters of the IDCT launches for single-type and mixed frames (one frame alone): tools/pmc_icache.sh [MIX ...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for T in "${@:-default DCT8}"; do
  n=$(echo $T | tr -c 'A-Za-z0-9' '_' | cut -c1-30)
  for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_IFETCH" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_IFETCH_LEVEL"; do
    s=$(echo $set | tr ' ' '_' | cut -c1-30)
    rocprofv3 --kernel-trace --output-format csv --pmc $set -d $R/gpurun_out/pmc_ic/$n/$s -o p -- python3 $R/tools/idct_mix_bench.py $T > /dev/null 2>&1
  done
  echo "== $T"
  cd $R && for f in $(find gpurun_out/pmc_ic/$n -name "*counter_collection.csv"); do python3 tools/pmc_summary.py $f | grep -A1 "k_idct"; done; cd /tmp
done

#!/bin/bash
# PMC A/B of a single-type IDCT frame between two library builds: tools/pmc_ab.sh TYPE lib1.so lib2.so ...
T=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for LIB in "$@"; do
  export JXL_AMD_LIB=$LIB
  echo "== $LIB"
  for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_VALU" "SQ_IFETCH SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" "SQ_INST_CYCLES_SMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_WAIT_ANY"; do
    rm -rf /tmp/pm && rocprofv3 --kernel-trace --output-format csv --pmc $set -d /tmp/pm -o p -- python3 $R/tools/idct_mix_bench.py $T > /dev/null 2>&1
    python3 $R/tools/pmc_summary.py $(find /tmp/pm -name "*counter_collection.csv") | grep -A1 "idct"
  done
done

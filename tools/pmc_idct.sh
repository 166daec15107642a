#!/bin/bash
# PMC profile of a single-type IDCT stage: tools/pmc_idct.sh DCT32   (run on the GPU box through gpurun)
T=${1:-DCT32}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAVES GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS" "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_LDS_BANK_CONFLICT" "SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_LDS_IDX_ACTIVE"; do
  n=$(echo $set | tr ' ' '_' | cut -c1-40)
  rocprofv3 --kernel-trace --output-format csv --pmc $set -d $R/gpurun_out/pmc_$T/$n -o p -- python3 $R/tools/idct_mix_bench.py $T > /dev/null 2>&1
done
cd $R && for f in $(find gpurun_out/pmc_$T -name "*counter_collection.csv"); do python3 tools/pmc_summary.py $f | grep -A1 "k_idct"; done

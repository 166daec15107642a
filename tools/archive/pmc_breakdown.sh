#!/bin/bash
# where do the waves of each kernel spend their cycles? three PMC passes over one 4K frame (GPU box): bash tools/pmc_breakdown.sh <tag> [bench args]
TAG=${1:-x}; shift
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/pmc_$TAG
cd /tmp && export TMPDIR=/tmp
ONE="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-gather --no-end-to-end --frames-per-gpu 1 $*"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA -d $OUT/a -o p -- $ONE > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_FLAT SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS_SMEM SQ_INST_CYCLES_SMEM -d $OUT/b -o p -- $ONE > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAIT_INST_LDS SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INSTS_BRANCH SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_VALU -d $OUT/c -o p -- $ONE > /dev/null 2>&1
python3 $ROOT/tools/valu_count.py $OUT

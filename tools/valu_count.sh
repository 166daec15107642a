#!/bin/bash
# dynamic instruction counts per kernel of one 4K frame (GPU box): bash tools/valu_count.sh <tag>   (environment knobs pass through)
TAG=${1:-x}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/valu_$TAG
cd /tmp && export TMPDIR=/tmp
ONE="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --frames-per-gpu 1"
rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES -d $OUT -o p -- $ONE > /dev/null 2>&1
python3 $ROOT/tools/valu_count.py $OUT

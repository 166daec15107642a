#!/bin/bash
# rocprofv3 passes behind profiles/rN_*: run on the GPU box through gpurun ("bash tools/profile_round.sh r1").
# Kernel-trace stats of the default bench, then PMC passes (separate runs, --kernel-trace only) of one frame alone.
R=${1:-r1}
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/prof_$R
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o p -- python3 $ROOT/bench.py --no-cpu-baseline --no-end-to-end --no-also > $OUT.bench_stats.log 2>&1
ONE="python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-end-to-end --no-also --frames-per-gpu 1"
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc FETCH_SIZE -d $OUT/fetch -o p -- $ONE > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc WRITE_SIZE -d $OUT/write -o p -- $ONE > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $OUT/sq -o p -- $ONE > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE SQ_INSTS_SALU -d $OUT/lds -o p -- $ONE > /dev/null 2>&1
find $OUT -name "*.csv" | head -30

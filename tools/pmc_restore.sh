cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/pmc_rest
rocprofv3 --kernel-trace --output-format csv --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY -d $R/gpurun_out/pmc_rest/sq -o p -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --frames-per-gpu 1 > /dev/null 2>&1
rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS -d $R/gpurun_out/pmc_rest/lds -o p -- python3 $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline --frames-per-gpu 1 > /dev/null 2>&1
cd $R && for f in $(find gpurun_out/pmc_rest -name "*counter_collection.csv"); do python3 tools/pmc_summary.py $f | grep -A1 "k_restore_fused"; done

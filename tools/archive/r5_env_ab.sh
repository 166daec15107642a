#!/bin/bash
# VarDCT env-switch A/B on one box (each switch read once per process): tools/r5_env_ab.sh "A=1 B=2" "-" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5; mkdir -p $OUT; cd $ROOT
for rep in 1 2; do
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  for fpg in 8 1; do
    env $e timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --no-gather --frames-per-gpu $fpg > $OUT/ab.json 2>$OUT/ab.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/ab.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("[%-40s fpg $fpg] value %6.0f Mpx/s  single %.4f ms | idct %.4f restore %.4f | in batch: idct %.4f restore %.4f" % ("$e", d["value"], d["config"].get("single_frame_ms",0), r["idct_stage_ms"], r["kernel_ms"], r["idct_stage_ms_in_batch"], r["kernel_ms_in_batch"]))
except Exception as e:
    print("[$e] bench failed", e); print(open("$OUT/ab.err").read()[-1500:])
PY
  done
done
done

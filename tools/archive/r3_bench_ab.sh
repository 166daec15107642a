#!/bin/bash
# A/B of the default bench in ONE gpurun call (same box): tools/r3_bench_ab.sh "ENV1=.. ENV2=.." "ENV=.." ...   (use - for no env)
mkdir -p gpurun_out/r3
i=0
for e in "$@"; do
  i=$((i+1))
  if [ "$e" = "-" ]; then e=""; fi
  env $e python bench.py --no-cpu-baseline --no-end-to-end --no-gather > gpurun_out/r3/ab_$i.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r3/ab_$i.json").read().strip().splitlines()[-1])
r=d["roofline"]
print("[$e] value %.0f Mpx/s  step %.4f ms  single %.4f ms | restore %.4f idct %.4f frame_ev %.4f | in batch: restore %.4f idct %.4f" % (d["value"], d["ms_per_step"], d["config"].get("single_frame_ms",0), r["kernel_ms"], r["idct_stage_ms"], r["frame_ms_events"], r["kernel_ms_in_batch"], r["idct_stage_ms_in_batch"]))
PY
done

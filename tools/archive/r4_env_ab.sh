#!/bin/bash
# A/B of environments inside ONE gpurun call: tools/r4_env_ab.sh "ENV=.." "-" ...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/r4
i=0
for e in "$@"; do
  i=$((i+1)); [ "$e" = "-" ] && e=""
  env $e python bench.py --no-cpu-baseline --no-end-to-end --no-gather > gpurun_out/r4/env_$i.json 2>/dev/null
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4/env_$i.json").read().strip().splitlines()[-1]); r=d["roofline"]
print("[%-28s] value %.0f Mpx/s  step %.4f ms  single %.4f ms | restore %.4f idct %.4f | in batch: restore %.4f idct %.4f" % ("$e", d["value"], d["ms_per_step"], d["config"].get("single_frame_ms",0), r["kernel_ms"], r["idct_stage_ms"], r["kernel_ms_in_batch"], r["idct_stage_ms_in_batch"]))
PY
done

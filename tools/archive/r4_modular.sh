#!/bin/bash
# Modular A/B on one box: tools/r4_modular.sh "ENV=.. ENV=.." ...   ("-" = defaults)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; mkdir -p gpurun_out/r4
for e in "$@"; do
  [ "$e" = "-" ] && e=""
  for wl in "modular1080p --frames-per-gpu 1" "modular1080p --frames-per-gpu 4" "modular8k --frames-per-gpu 1" "modular8k --frames-per-gpu 4"; do
    env $e python bench.py --workload $wl --no-cpu-baseline --no-gather --no-end-to-end > gpurun_out/r4/mod.json 2>gpurun_out/r4/mod.err
    python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/r4/mod.json").read().strip().splitlines()[-1])
    print("[%-48s] %-34s value %8.0f Mpx/s  ms_per_step %.4f  launches %s" % ("$e", "$wl", d["value"], d["ms_per_step"], d["config"].get("kernel_launches_per_frame", d["config"].get("launches_per_image"))))
except Exception as e:
    print("[$e] $wl failed", e); print(open("gpurun_out/r4/mod.err").read()[-800:])
PY
  done
done

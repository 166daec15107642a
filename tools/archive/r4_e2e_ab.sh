#!/bin/bash
# the default bench line (with its end-to-end legs) for the product library and tagged builds, in ONE gpurun call: tools/r4_e2e_ab.sh tag...
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
cd $ROOT
for t in - "$@" -; do
  if [ "$t" = "-" ]; then unset JXL_AMD_LIB; name=product; else export JXL_AMD_LIB=$ROOT/jxlatte_amd/libjxlatte_amd_$t.so; name=$t; fi
  timeout 900 python bench.py --no-cpu-baseline > $OUT/e2e_$name.json 2>$OUT/e2e_$name.err
  python - <<PY
import json
d=json.loads(open("$OUT/e2e_$name.json").read().strip().splitlines()[-1])
e=d.get("untimed") or {}
print("[$name] value %.0f" % d["value"], json.dumps(e)[:1500])
PY
done

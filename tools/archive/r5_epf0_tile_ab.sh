#!/bin/bash
# window height of the iteration-0 kernel (tagged builds): bash tools/r5_epf0_tile_ab.sh
O=gpurun_out/r5_epf0_tile_ab.txt; : > $O
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for t in ${TAGS:-- wh32 wh48}; do
  if [ "$t" = "-" ]; then unset JXL_AMD_LIB; else export JXL_AMD_LIB=$ROOT/jxlatte_amd/libjxlatte_amd_$t.so; fi
  for spec in "batch --epf-iters 3" "single --epf-iters 3 --frames-per-gpu 1"; do
    set -- $spec; name=$1; shift
    timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --no-gather --verify "$@" > /tmp/s.json 2>/tmp/s.err
    python -c "
import json
try:
    d=json.loads(open('/tmp/s.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('lib %-5s %-7s value %8.1f ms/step %.4f idct %s restore %s' % ('$t', '$name', d['value'], d['ms_per_step'], r.get('idct_stage_ms'), r.get('kernel_ms')))
except Exception as ex: print('$t $name failed', ex, open('/tmp/s.err').read()[-600:])" >> $O
  done
done
done
cat $O

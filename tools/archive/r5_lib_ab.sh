#!/bin/bash
# generic A/B of tagged library builds: TAGS="- nt" bash tools/r5_lib_ab.sh  ("-" = product); lines: batch, single, saturated IDCT stage
O=gpurun_out/r5_lib_ab.txt; : > $O
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for rep in 1 2; do
for t in ${TAGS:--}; do
  if [ "$t" = "-" ]; then unset JXL_AMD_LIB; else export JXL_AMD_LIB=$ROOT/jxlatte_amd/libjxlatte_amd_$t.so; fi
  for spec in "default 8 15" "default 1 15" "default 8 1" "dct8 8 1"; do
    set -- $spec; VER=--verify; [ "$3" = "1" ] && VER=""
    timeout 300 python bench.py --stages $3 --mix $1 --frames-per-gpu $2 --no-cpu-baseline --no-end-to-end --no-gather $VER > /tmp/s.json 2>/tmp/s.err
    python -c "
import json
try:
    d=json.loads(open('/tmp/s.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('lib %-5s mix %-7s N=%s stages %-2s: value %7.0f ms/step %.4f per frame %.4f' % ('$t', '$1', '$2', '$3', d['value'], d['ms_per_step'], d['ms_per_step']/$2))
except Exception as e: print('$t $spec failed', e, open('/tmp/s.err').read()[-300:])" >> $O
  done
done
done
sort $O

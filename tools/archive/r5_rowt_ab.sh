#!/bin/bash
# row pass of 8-wide blocks: lanes -> blocks of one row (product) against lanes -> rows of one block (-DJXL_WG3_ROW_T=0, tag rt0)
O=gpurun_out/r5_rowt_ab.txt; : > $O
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
[ -n "$SKIPTESTS" ] || python -m pytest tests -m gpu -x -q -k "frame_parity or idct or 4k or fuzz or large or sub" 2>&1 | tail -1 >> $O
for rep in 1 2; do
for t in - rt0; do
  if [ "$t" = "-" ]; then unset JXL_AMD_LIB; else export JXL_AMD_LIB=$ROOT/jxlatte_amd/libjxlatte_amd_$t.so; fi
  for spec in "default 8 15" "default 1 15" "default 8 1" "dct8 8 1" "dct8 1 15" "all 8 15"; do
    set -- $spec; VER=--verify; [ "$3" = "1" ] && VER=""
    timeout 300 python bench.py --stages $3 --mix $1 --frames-per-gpu $2 --no-cpu-baseline --no-end-to-end --no-gather $VER > /tmp/s.json 2>/tmp/s.err
    python -c "
import json
try:
    d=json.loads(open('/tmp/s.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('lib %-4s mix %-7s N=%s stages %-2s: value %7.0f ms/step %.4f per frame %.4f idct_alone %s' % ('$t', '$1', '$2', '$3', d['value'], d['ms_per_step'], d['ms_per_step']/$2, r.get('idct_stage_ms')))
except Exception as e: print('$t $spec failed', e, open('/tmp/s.err').read()[-400:])" >> $O
  done
done
done
cat $O

#!/bin/bash
# timing-only ablations of k_idct_wg3 (tagged builds, wrong results) alone and with 8 frames in flight: what does the SATURATED IDCT stage wait for?
O=gpurun_out/r5_idct_ablation.txt; : > $O
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
for t in - abl_NOSTORE abl_NOLOAD abl_ALL abl_NOMAC abl_NOBAR; do
  if [ "$t" = "-" ]; then unset JXL_AMD_LIB; else export JXL_AMD_LIB=$ROOT/jxlatte_amd/libjxlatte_amd_$t.so; fi
  for mix in default dct8; do
    line="lib $t mix $mix:"
    for f in 1 8; do
      timeout 300 python bench.py --stages 1 --mix $mix --frames-per-gpu $f --no-cpu-baseline --no-end-to-end --no-gather > /tmp/s.json 2>/tmp/s.err
      v=$(python -c "
import json
try:
    d=json.loads(open('/tmp/s.json').read().strip().splitlines()[-1]); print('%.4f' % (d['ms_per_step']/$f))
except Exception as e: print('fail')")
      line="$line  N=$f $v ms/frame"
    done
    echo "$line" >> $O
  done
done
cat $O

#!/bin/bash
# three EPF iterations as one launch (JXL_EPF3_SPLIT=0) or as two: parity tests, then the EPF x 3 bench lines on one box
O=gpurun_out/r5_epf3_split_ab.txt; : > $O
python -m pytest tests -m gpu -x -q -k "epf or restore or fuzz or frame_parity or 4k or pq or post or planes or decode" 2>&1 | tail -2 >> $O
for rep in 1 2; do
for e in 0 1; do
  for spec in "batch --epf-iters 3" "single --epf-iters 3 --frames-per-gpu 1" "rgb8 --epf-iters 3 --frames-per-gpu 1 --no-gather"; do
    set -- $spec; name=$1; shift
    JXL_EPF3_SPLIT=$e timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --no-gather --verify "$@" > /tmp/s.json 2>/tmp/s.err
    python -c "
import json
try:
    d=json.loads(open('/tmp/s.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('EPF3_SPLIT=$e %-7s value %8.1f ms/step %.4f idct %s restore %s' % ('$name', d['value'], d['ms_per_step'], r.get('idct_stage_ms'), r.get('kernel_ms')))
except Exception as ex: print('$e $name failed', ex, open('/tmp/s.err').read()[-600:])" >> $O
  done
done
done
cat $O

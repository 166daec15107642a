#!/bin/bash
# how many frames should be in flight: --stream-groups G x GPU_MAX_HW_QUEUES, one box
O=gpurun_out/r5_groups_ab.txt; : > $O
for rep in 1 2; do
for spec in "4 0" "4 2" "4 3" "4 4" "16 0" "16 2" "16 3" "16 4" "8 2" "8 4" "12 3"; do
  set -- $spec
  GPU_MAX_HW_QUEUES=$1 timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --no-gather --stream-groups $2 > /tmp/s.json 2>/tmp/s.err
  python -c "
import json
try:
    d=json.loads(open('/tmp/s.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('queues %-3s groups %s  value %8.1f ms/step %.4f' % ('$1', '$2', d['value'], d['ms_per_step']))
except Exception as ex: print('$spec failed', ex)" >> $O
done
done
sort $O

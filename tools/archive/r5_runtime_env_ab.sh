#!/bin/bash
# runtime settings against the default timed step (8 x 4K batch + single frame), one box: bash tools/r5_runtime_env_ab.sh
O=gpurun_out/r5_runtime_env_ab.txt; : > $O
for rep in 1 2; do
for e in "A=1" "GPU_MAX_HW_QUEUES=2" "GPU_MAX_HW_QUEUES=6" "GPU_MAX_HW_QUEUES=8" "HIP_FORCE_DEV_KERNARG=1" "HIP_FORCE_DEV_KERNARG=0" "HSA_ENABLE_INTERRUPT=0" "GPU_STREAMOPS_CP_WAIT=1" "DEBUG_CLR_GRAPH_PACKET_CAPTURE=1"; do
  env $e timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --no-gather > /tmp/s.json 2>/tmp/s.err
  python -c "
import json
try:
    d=json.loads(open('/tmp/s.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
    print('%-36s value %8.1f ms/step %.4f single %.4f idct %s restore %s' % ('$e', d['value'], d['ms_per_step'], d['config'].get('single_frame_ms',0), r.get('idct_stage_ms'), r.get('kernel_ms')))
except Exception as ex: print('$e failed', ex)" >> $O
done
done
sort $O

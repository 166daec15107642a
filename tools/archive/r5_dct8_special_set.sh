#!/bin/bash
# JXL_DCT8_SPECIAL=0/1 over the round's bench lines (one box): bash tools/r5_dct8_special_set.sh
O=gpurun_out/r5_dct8_special_set.txt; : > $O
L=tests/golden/samples_large
one() { for e in 0 1 0 1; do JXL_DCT8_SPECIAL=$e timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --no-gather "${@:2}" > /tmp/s.json 2>/tmp/s.err; python -c "
import json
d=json.loads(open('/tmp/s.json').read().strip().splitlines()[-1]); r=d.get('roofline',{})
print('%-22s DCT8_SPECIAL=$e value %8.1f ms/step %.4f idct %s restore %s' % ('$1', d['value'], d['ms_per_step'], r.get('idct_stage_ms'), r.get('kernel_ms')))" >> $O; done; }
one vardct4k
one vardct4k_single --frames-per-gpu 1
one vardct4k_epf3 --epf-iters 3
one vardct8k_pq --workload vardct8k_pq --frames-per-gpu 2
one sollevante4k --workload jxlfile --input $L/sollevante-hdr.jxl
one bbb720p_batch --workload jxlfile --input tests/golden/samples/bbb.jxl --batch
one mix_all --mix all
cat $O

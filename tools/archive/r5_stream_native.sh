#!/bin/bash
# streaming boundary leg: native host threads (tools/native/stream_bench.cpp) against Python threads, and a sweep of the context count
# (GPU box): bash tools/r5_stream_native.sh
O=gpurun_out/r5_stream_native.txt
: > $O
show() { python -c "
import json,sys
d=json.loads(open('/tmp/sb.json').read().strip().splitlines()[-1]); s=d['untimed']['streaming']
print('$1', {k:v for k,v in s.items() if k not in ('note','python_threads')})
if 'python_threads' in s: print('   python threads:', {k:v for k,v in s['python_threads'].items() if k!='note'})
" >> $O; }
JXL_BENCH_STREAM_BOTH=1 timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /tmp/sb.json 2>/tmp/sb.err; show "ctx 8 both"
for n in 4 8 12 16 24; do
  JXL_BENCH_STREAM_CTX=$n timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /tmp/sb.json 2>/tmp/sb.err; show "ctx $n"
done
cat $O

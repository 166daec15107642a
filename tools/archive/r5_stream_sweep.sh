#!/bin/bash
# native streaming leg under a list of environments, in ONE gpurun call (boxes differ): bash tools/r5_stream_sweep.sh "N ENV=.. ENV=.." ...
O=gpurun_out/r5_stream_sweep.txt
: > $O
for rep in 1 2; do
for spec in "$@"; do
  set -- $spec
  n=$1; shift
  line=$(env "$@" timeout 120 python3 tools/r5_stream.py $n ${FPC:-24} 2>&1 | tail -1 | python3 -c "
import sys,ast
try:
    d=ast.literal_eval(sys.stdin.read().strip()); print('%.3f ms/frame %7.1f Mpx/s same=%s %s' % (d['ms_per_frame'], d['streaming_end_to_end_Mpx_s'], d['identical_output'], d['host_ms_per_frame_and_thread']))
except Exception as e: print('failed', e)")
  echo "$spec :: $line" >> $O
done
done
sort $O

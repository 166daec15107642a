#!/bin/bash
# shader / memory clock and power while the default batch runs (rocm-smi polled beside a long bench run)
python bench.py --steps 4000 --warmup 20 --no-cpu-baseline --no-end-to-end --no-gather > /tmp/cw.json 2>/dev/null &
BP=$!
sleep 6
for i in 1 2 3 4 5 6 7 8; do
  rocm-smi --showclocks --showpower --showuse 2>/dev/null | grep -E "sclk|mclk|fclk|Power|GPU use" | tr '\n' ' '; echo
  sleep 0.5
done
wait $BP
python -c "
import json
d=json.loads(open('/tmp/cw.json').read().strip().splitlines()[-1]); print('bench', d['value'], d['ms_per_step'])"
echo idle:
rocm-smi --showclocks --showpower 2>/dev/null | grep -E "sclk|mclk|Power" | tr '\n' ' '; echo

#!/bin/bash
# where prepare's host time goes with N contexts in flight (JXL_PREPARE_TIMING sections, mean ms): bash tools/r5_stream_sections.sh N [ENV=..]
N=${1:-8}; shift
env "$@" JXL_PREPARE_TIMING=1 timeout 120 python3 tools/r5_stream.py $N 8 2> /tmp/sec.err | tail -1 | cut -c1-400
python3 - <<'PY'
import collections, re
acc = collections.defaultdict(list)
for l in open('/tmp/sec.err'):
    m = re.match(r'\[(\w+)\] (.+?)\s+([\d.]+) ms', l)
    if m: acc[(m.group(1), m.group(2).strip())].append(float(m.group(3)))
for k, v in acc.items():
    v2 = v[len(v) // 4:]
    print("  %-10s %-30s n %4d  mean %.3f ms  max %.3f" % (k[0], k[1], len(v2), sum(v2) / len(v2), max(v2)))
PY

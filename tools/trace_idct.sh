#!/bin/bash
# kernel-trace of single-type IDCT frames: tools/trace_idct.sh [lib.so] TYPE...   (GPU box)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
LIB=$1; shift
export JXL_AMD_LIB=$LIB
rm -rf /tmp/tr && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/tr -o p -- python3 $R/tools/idct_mix_bench.py "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/tr/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'idct' in r['Name']: print('%-60s calls %s avg %.1f us' % (r['Name'][:60], r['Calls'], float(r['AverageNs'])/1e3))
PY

#!/bin/bash
# kernel timeline of one batch step: concurrency and idle gaps (GPU box): tools/timeline.sh [stages] [frames]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python3 $R/tools/diag_submit.py ${1:-31} ${2:-8} > /tmp/tl.log 2>&1
tail -1 /tmp/tl.log
python3 - <<PY
import csv, glob
f = glob.glob('/tmp/tl/**/*kernel_trace.csv', recursive=True)[0]
rows = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void jxl::','').replace('(anonymous namespace)::','')[:28], r.get('Queue_Id','?')) for r in csv.DictReader(open(f))]
rows.sort()
# take the last 1/4 of the run (steady state), one step = 8 frames
n = len(rows)
per_step = None
names = [r[2] for r in rows]
sub = rows[-(n // 6):]
t0 = sub[0][0]
ev = []
for s, e, k, q in sub: ev += [(s, 1), (e, -1)]
ev.sort()
busy = 0; conc_time = {}; cur = 0; last = ev[0][0]
for t, d in ev:
    conc_time[cur] = conc_time.get(cur, 0) + (t - last)
    last = t; cur += d
tot = ev[-1][0] - ev[0][0]
print('window %.1f us, %d kernels' % (tot / 1e3, len(sub)))
for c in sorted(conc_time): print('  %d kernels in flight: %5.1f %%' % (c, 100.0 * conc_time[c] / tot))
import collections
d = collections.defaultdict(list)
for s, e, k, q in sub: d[k].append((e - s) / 1e3)
for k, v in d.items(): print('  %-30s n=%3d avg %.1f us' % (k, len(v), sum(v) / len(v)))
for s, e, k, q in sub[:40]: print('   %8.1f %8.1f  %-28s q=%s' % ((s - t0) / 1e3, (e - t0) / 1e3, k, q))
PY

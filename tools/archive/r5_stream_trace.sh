#!/bin/bash
# device timeline of the native streaming leg: which kernels overlap (GPU box): bash tools/r5_stream_trace.sh [n_ctx] [ENV=..]
ROOT=$GRAFT_REPO_ROOT; N=${1:-8}; shift
for e in "$@"; do export "$e"; done
cd /tmp && export TMPDIR=/tmp
python3 $ROOT/tools/r5_stream.py $N 8 2>&1 | tail -1
rm -rf /tmp/st && timeout 300 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/st -o p -- python3 $ROOT/tools/r5_stream.py $N 8 > /tmp/st.log 2>&1
tail -1 /tmp/st.log
python3 - <<'PY'
import csv, glob, collections
kt = [r for f in glob.glob('/tmp/st/**/*kernel_trace.csv', recursive=True) for r in csv.DictReader(open(f))]
cp = [r for f in glob.glob('/tmp/st/**/*memory_copy_trace.csv', recursive=True) for r in csv.DictReader(open(f))]
import re
kn = lambda n: (re.search(r'(k_\w+)', n) or [n[:40]])[0] if re.search(r'(k_\w+)', n) else n[:40]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), kn(r['Kernel_Name'])) for r in kt]
ev += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '?')) for r in cp]
t0 = min(e[0] for e in ev); t1 = max(e[1] for e in ev)
lo = sorted(e[0] for e in ev)[len(ev) // 3]  # steady state: from a third of the launches on
def busy(iv):
    iv = sorted((max(a, lo), b) for a, b in iv if b > lo)
    tot = 0; ce = lo
    for a, b in iv:
        if b <= ce: continue
        tot += b - max(a, ce); ce = b
    return tot
span = t1 - lo
print("steady window %.1f ms" % (span / 1e6))
by = collections.defaultdict(list)
for a, b, k in ev: by[k].append((a, b))
for k, iv in sorted(by.items(), key=lambda x: -busy(x[1]))[:12]:
    n = sum(1 for a, b in iv if b > lo)
    print("  %-42s busy %5.1f %%  n %4d  mean %.3f ms" % (k, 100.0 * busy(iv) / span, n, sum(b - a for a, b in iv if b > lo) / max(n, 1) / 1e6))
pc = [(a, b) for a, b, k in ev if k in ('k_widen2d_host8', 'k_copy16') or k.startswith('COPY')]
cm = [(a, b) for a, b, k in ev if not (k in ('k_widen2d_host8', 'k_copy16') or k.startswith('COPY'))]
print("  a bus transfer (widen / copy16 / runtime copy) in flight %5.1f %%; a compute kernel in flight %5.1f %%" % (100.0 * busy(pc) / span, 100.0 * busy(cm) / span))
nfr = sum(1 for a, b, k in ev if k == 'k_widen2d_host8' and b > lo) / 3.0
print("  frames in the window: %.1f -> %.3f ms per frame" % (nfr, span / 1e6 / max(nfr, 1)))
print("  anything running %5.1f %%" % (100.0 * busy([(a, b) for a, b, k in ev]) / span))
# concurrency histogram
pts = sorted([(max(a, lo), 1) for a, b, k in ev if b > lo] + [(b, -1) for a, b, k in ev if b > lo])
cur = 0; last = lo; hist = collections.Counter()
for t, dlt in pts:
    hist[cur] += t - last; last = t; cur += dlt
print("  concurrency (kernels+copies in flight): " + ", ".join("%d: %.0f%%" % (k, 100.0 * v / span) for k, v in sorted(hist.items())))
PY
python3 - <<'PY'
# timeline of the steady state: every launch of 5 ms, one line each, with its queue
import csv, glob
kt = [r for f in glob.glob('/tmp/st/**/*kernel_trace.csv', recursive=True) for r in csv.DictReader(open(f))]
cp = [r for f in glob.glob('/tmp/st/**/*memory_copy_trace.csv', recursive=True) for r in csv.DictReader(open(f))]
ev = [(int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('void ', '').replace('jxl::', '').replace('(anonymous namespace)::', '')[:28], r.get('Queue_Id', '?')) for r in kt]
ev += [(int(r['Start_Timestamp']), int(r['End_Timestamp']), 'COPY ' + r.get('Direction', '?')[12:], 'dma') for r in cp]
ev.sort()
t0 = ev[0][0]; t1 = ev[-1][1]
lo = sorted(e[0] for e in ev)[len(ev) * 6 // 10]
print("timeline from +%.1f ms (start ms, duration ms, queue, name)" % ((lo - t0) / 1e6))
for a, b, k, q in ev:
    if a >= lo and a < lo + 7_000_000:
        print("  %8.3f %7.3f  q%-4s %s" % ((a - lo) / 1e6, (b - a) / 1e6, q, k))
PY

#!/bin/bash
# DMA / kernel occupancy of the streaming leg: bash tools/r4_stream_trace.sh [n_ctx]
ROOT=$GRAFT_REPO_ROOT; N=${1:-12}
cd /tmp && export TMPDIR=/tmp
for n in 1 4 $N; do python3 $ROOT/tools/r4_stream.py $n 8 2>&1 | tail -2; done
rm -rf /tmp/st && rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/st -o p -- python3 $ROOT/tools/r4_stream.py $N 8 > /tmp/st.log 2>&1
tail -2 /tmp/st.log
python3 - <<'PY'
import csv, glob, collections
cp = [r for f in glob.glob('/tmp/st/**/*memory_copy_trace.csv', recursive=True) for r in csv.DictReader(open(f))]
kt = [r for f in glob.glob('/tmp/st/**/*kernel_trace.csv', recursive=True) for r in csv.DictReader(open(f))]
if not cp:
    print("no copy trace"); raise SystemExit
t0 = min(int(r['Start_Timestamp']) for r in cp + kt); t1 = max(int(r['End_Timestamp']) for r in cp + kt)
# the last 60 % of the run (steady state)
lo = t0 + (t1 - t0) * 4 // 10
def busy(iv):
    iv = sorted((max(a, lo), b) for a, b in iv if b > lo)
    tot = 0; ce = lo
    for a, b in iv:
        if b <= ce: continue
        tot += b - max(a, ce); ce = b
    return tot
by = collections.defaultdict(list); sz = collections.defaultdict(int)
for r in cp:
    k = r.get('Direction', r.get('Kind', '?'))
    by[k].append((int(r['Start_Timestamp']), int(r['End_Timestamp'])))
    if int(r['End_Timestamp']) > lo: sz[k] += int(r.get('Size', r.get('Bytes', 0)) or 0)
span = t1 - lo
print("steady window %.1f ms" % (span / 1e6))
for k, iv in by.items():
    b = busy(iv)
    print("  copies %-28s busy %5.1f %%  %8.1f MB  -> %.1f GB/s while busy, %d copies" % (k, 100.0 * b / span, sz[k] / 1e6, sz[k] / max(b, 1), len(iv)))
kb = busy([(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in kt])
print("  kernels: some kernel running %5.1f %% of the window" % (100.0 * kb / span))
allb = busy([(int(r['Start_Timestamp']), int(r['End_Timestamp'])) for r in kt + cp])
print("  device doing anything (kernel or copy) %5.1f %%" % (100.0 * allb / span))
PY

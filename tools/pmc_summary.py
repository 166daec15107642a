#!/usr/bin/env python3
"""summarise rocprofv3 --pmc counter_collection.csv per kernel (mean over dispatches)"""
import csv, sys, collections
for path in sys.argv[1:]:
    rows = list(csv.DictReader(open(path)))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        acc[r['Kernel_Name'][:60]][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in acc.items():
        if 'rocclr' in k: continue
        print(k)
        print('   ' + '  '.join('%s=%.3g' % (n, sum(v) / len(v)) for n, v in sorted(d.items())))

"""mean counter value per kernel from a rocprofv3 --pmc csv directory (tools/valu_count.sh)"""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"].split("(")[0][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
tot = collections.Counter()
for k, d in sorted(acc.items()):
    n = len(next(iter(d.values())))
    if n < 3:
        continue
    print("%-62s n=%-3d " % (k, n) + "  ".join("%s %.3e" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
    for c, v in d.items():
        tot[c] += sum(v) / len(v)
print("sum over kernels (one launch each): " + "  ".join("%s %.3e" % kv for kv in sorted(tot.items())))

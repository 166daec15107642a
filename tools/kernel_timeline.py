"""print the per-kernel timeline (start offset, duration) of the last N dispatches of a rocprofv3 --kernel-trace CSV:
    python tools/kernel_timeline.py <dir-or-csv> [n_last]"""
import csv, glob, os, sys
p = sys.argv[1]
n_last = int(sys.argv[2]) if len(sys.argv) > 2 else 12
files = [p] if p.endswith(".csv") else glob.glob(os.path.join(p, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:60], r.get("Stream_Id", r.get("Queue_Id", "?")),
                     r.get("VGPR_Count", "?"), r.get("Grid_Size", r.get("Workgroup_Size", "?"))))
rows.sort()
rows = rows[-n_last:]
t0 = rows[0][0]
for s, e, name, q, vg, grid in rows:
    print("%9.1f us  +%7.1f us  q=%s vgpr=%s grid=%s  %s" % ((s - t0) / 1e3, (e - s) / 1e3, q, vg, grid, name))
print("span %.1f us" % ((max(r[1] for r in rows) - t0) / 1e3))

#!/bin/bash
# Dynamic VALU instruction count of the restoration kernel per stage (GPU box): the same 4K frame with stages switched off one at a
# time, SQ_INSTS_VALU of the k_restore_fused launch from a rocprofv3 --pmc pass each; differences = the stages' own instructions
# (incl. the halo they add to the stages in front of them).   bash tools/r3_stage_split.sh > gpurun_out/r3_stage_split.txt
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() {  # label, bench args
  rm -rf /tmp/ss
  rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES -d /tmp/ss -o p -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-gather --no-end-to-end --frames-per-gpu 1 "${@:2}" > /dev/null 2>&1
  python3 - "$1" <<PY
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob('/tmp/ss/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'k_restore_fused' in r['Kernel_Name'] or 'k_gab' in r['Kernel_Name'] or 'k_xyb' in r['Kernel_Name'] or 'k_epf' in r['Kernel_Name']:
            acc[r['Kernel_Name'].split('(')[0][-60:]][r['Counter_Name']].append(float(r['Counter_Value']))
for k, d in acc.items():
    print("%-34s %-46s " % (sys.argv[1], k) + "  ".join("%s %.4e" % (c, sum(v) / len(v)) for c, v in sorted(d.items())))
PY
}
run "Gab+EPF1+EPF2+XYB (stages 31)" --stages 31
run "Gab+EPF1+EPF2 (stages 7)" --stages 7
run "Gab+EPF1 (stages 7, iters 1)" --stages 7 --epf-iters 1
run "Gab (stages 3)" --stages 3
run "EPF1+EPF2+XYB (stages 13)" --stages 13
run "EPF1+EPF2 (stages 5)" --stages 5
run "EPF1 (stages 5, iters 1)" --stages 5 --epf-iters 1
run "XYB only (stages 9)" --stages 9
run "Gab+EPF0+EPF1+EPF2+XYB (iters 3)" --stages 31 --epf-iters 3

#!/bin/bash
# A/B of tagged library builds in ONE gpurun call (same box): tools/r4_ab.sh [--pmc] [--tests] tag1 tag2 ...   ("-" = the product library)
# per tag: the default bench line (batch + single frame + stage times); with --pmc also SQ_INSTS_VALU / LDS / wave cycles of one frame
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
PMC=0; TESTS=0
while [ "${1:0:2}" = "--" ]; do
  [ "$1" = "--pmc" ] && PMC=1
  [ "$1" = "--tests" ] && TESTS=1
  shift
done
cd $ROOT
if [ $TESTS = 1 ]; then
  timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
fi
for t in "$@"; do
  if [ "$t" = "-" ]; then unset JXL_AMD_LIB; name=product; else export JXL_AMD_LIB=$ROOT/jxlatte_amd/libjxlatte_amd_$t.so; name=$t; fi
  cd $ROOT
  timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --no-gather > $OUT/ab_$name.json 2>$OUT/ab_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/ab_$name.json").read().strip().splitlines()[-1])
    r=d["roofline"]
    print("[$name] value %.0f Mpx/s  step %.4f ms  single %.4f ms | restore %.4f idct %.4f frame_ev %.4f | in batch: restore %.4f idct %.4f" % (d["value"], d["ms_per_step"], d["config"].get("single_frame_ms",0), r["kernel_ms"], r["idct_stage_ms"], r["frame_ms_events"], r["kernel_ms_in_batch"], r["idct_stage_ms_in_batch"]))
except Exception as e:
    print("[$name] bench failed", e); print(open("$OUT/ab_$name.err").read()[-2000:])
PY
  if [ $PMC = 1 ]; then
    cd /tmp && export TMPDIR=/tmp
    rm -rf $OUT/v_$name
    rocprofv3 --kernel-trace --output-format csv --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_SALU -d $OUT/v_$name -o p -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --frames-per-gpu 1 --no-end-to-end --no-gather > $OUT/v_$name.log 2>&1
    rocprofv3 --kernel-trace --output-format csv --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAVES -d $OUT/v_$name/lds -o p -- python3 $ROOT/bench.py --steps 5 --warmup 1 --no-cpu-baseline --frames-per-gpu 1 --no-end-to-end --no-gather > $OUT/v_$name.lds.log 2>&1
    python3 - <<PY
import csv,collections,glob
acc=collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/v_$name/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        acc[r['Kernel_Name'][:70]][r['Counter_Name']].append(float(r['Counter_Value']))
dur=collections.defaultdict(list)
for f in glob.glob("$OUT/v_$name/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        dur[r['Kernel_Name'][:70]].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1000)
for k,v in acc.items():
    if 'restore' in k:
        d=sorted(dur[k]); print("  %-72s us %.1f  %s" % (k, d[len(d)//2] if d else -1, {c:'%.4g'%(sum(x)/len(x)) for c,x in v.items()}))
PY
  fi
done

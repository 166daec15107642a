#!/bin/bash
# IDCT stage A/B of tagged library builds on one box: tools/r5_idct_ab.sh tag1 tag2 ...  ("-" = the product library)
# per tag: the default batch line + single-frame stage times for the default mix and for a frame of 8x8 DCT blocks only
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out/r5; mkdir -p $OUT; cd $ROOT
for rep in 1 2; do
for t in "$@"; do
  if [ "$t" = "-" ]; then unset JXL_AMD_LIB; name=product; else export JXL_AMD_LIB=$ROOT/jxlatte_amd/libjxlatte_amd_$t.so; name=$t; fi
  for mix in default dct8; do
    fpg=8; [ $mix = dct8 ] && fpg=1
    timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --no-gather --mix $mix --frames-per-gpu $fpg > $OUT/ab.json 2>$OUT/ab.err
    python - <<PY
import json
try:
    d=json.loads(open("$OUT/ab.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("[%-8s %-7s fpg $fpg] value %6.0f Mpx/s  single %.4f ms | idct %.4f restore %.4f | in batch: idct %.4f restore %.4f" % ("$name", "$mix", d["value"], d["config"].get("single_frame_ms",0), r["idct_stage_ms"], r["kernel_ms"], r["idct_stage_ms_in_batch"], r["kernel_ms_in_batch"]))
except Exception as e:
    print("[$name $mix] bench failed", e); print(open("$OUT/ab.err").read()[-1500:])
PY
  done
done
done

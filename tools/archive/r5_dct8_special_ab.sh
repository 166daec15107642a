#!/bin/bash
# A/B: 8x8 DCT blocks through k_idct_special_wg (JXL_DCT8_SPECIAL=1) against the persistent kernel, default mix (8 frames) and a frame
# of 8x8 DCT blocks only: bash tools/r5_dct8_special_ab.sh [ENV=.. for both sides]
O=gpurun_out/r5_dct8_special_ab.txt; : > $O
for rep in 1 2; do
for e in 0 1; do
  for mix in default dct8; do
    fpg=8; [ $mix = dct8 ] && fpg=1
    env "$@" JXL_DCT8_SPECIAL=$e timeout 600 python bench.py --no-cpu-baseline --no-end-to-end --no-gather --verify --mix $mix --frames-per-gpu $fpg > /tmp/ab.json 2>/tmp/ab.err
    python - >> $O <<PY
import json
try:
    d=json.loads(open("/tmp/ab.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("[DCT8_SPECIAL=$e %-7s fpg $fpg] value %6.0f Mpx/s  single %.4f ms | idct %.4f restore %.4f | in batch: idct %.4f restore %.4f" % ("$mix", d["value"], d["config"].get("single_frame_ms",0), r["idct_stage_ms"], r["kernel_ms"], r["idct_stage_ms_in_batch"], r["kernel_ms_in_batch"]))
except Exception as e:
    print("[$e $mix] bench failed", e); print(open("/tmp/ab.err").read()[-1500:])
PY
  done
done
done
cat $O

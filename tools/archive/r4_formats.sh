#!/bin/bash
# same-box A/B of the output-format variants: tools/r4_formats.sh tag...   ("-" = product)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/r4
mkdir -p $OUT
cd $ROOT
for t in "$@"; do
  if [ "$t" = "-" ]; then unset JXL_AMD_LIB; name=product; else export JXL_AMD_LIB=$ROOT/jxlatte_amd/libjxlatte_amd_$t.so; name=$t; fi
  python bench.py --workload vardct8k_pq --no-cpu-baseline --no-end-to-end --no-gather > $OUT/pq_$name.json 2>$OUT/pq_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("$OUT/pq_$name.json").read().strip().splitlines()[-1]); r=d["roofline"]
    print("[$name] 8K PQ-u16: value %.0f Mpx/s step %.4f ms single %.4f ms | restore %.4f idct %.4f path_frac %s" % (d["value"], d["ms_per_step"], d["config"].get("single_frame_ms",0), r["kernel_ms"], r["idct_stage_ms"], r.get("path_frac")))
except Exception as e:
    print("[$name] failed", e); print(open("$OUT/pq_$name.err").read()[-1500:])
PY
  python - <<PY
import sys, time
sys.path.insert(0, "$ROOT")
import numpy as np, ctypes as C
from jxlatte_amd import abi, host, synth, _lib
ctx = _lib.Context(0)
frame = synth.make_vardct_frame(3840, 2160, seed=1234, mix="default")
for name, tr, fmt in (("f32", abi.TRANSFER_NONE, abi.OUT_F32), ("srgb_rgb8", abi.TRANSFER_SRGB, abi.OUT_RGB8), ("srgb_rgb16", abi.TRANSFER_SRGB, abi.OUT_RGB16), ("pq_rgb16", abi.TRANSFER_PQ, abi.OUT_RGB16), ("pq_u16", abi.TRANSFER_PQ, abi.OUT_U16), ("srgb_u8", abi.TRANSFER_SRGB, abi.OUT_U8)):
    pp = abi.VarDCTParams.from_buffer_copy(frame["params"])
    pp.transfer, pp.out_format, pp.stages = tr, fmt, 31
    f2 = dict(frame); f2["params"] = bytes(pp)
    fr = host.Frame.from_synth(ctx, f2)
    for _ in range(3): fr.run()
    ctx.call("jxl_vardct_enable_stage_timing", 1)
    for _ in range(20): fr.run()
    v = C.c_float(); ctx.call("jxl_vardct_last_stage_ms", 2, C.byref(v))
    ctx.call("jxl_vardct_enable_stage_timing", 0)
    print("[$name] 4K %-11s restoration kernel %.1f us" % (name, v.value * 1e3))
PY
done

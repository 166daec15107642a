cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4
for n in 4 6 8 12; do
  JXL_BENCH_STREAM_CTX=$n python bench.py --no-cpu-baseline --no-gather --steps 5 > gpurun_out/r4/sctx.json 2>gpurun_out/r4/sctx.err
  python - <<PY
import json
d=json.loads(open("gpurun_out/r4/sctx.json").read().strip().splitlines()[-1])
s=d["untimed"]["streaming"]; print("ctx $n:", s.get("streaming_end_to_end_Mpx_s"), s.get("ms_per_frame"), "mapped e2e", d["untimed"]["mapped_i16"]["end_to_end_Mpx_s"])
PY
done
python3 tools/r4_stream.py 8 8 | tail -2

for st in 1 30 31; do for aux in 0 1 3; do
echo -n "stages=$st aux=$aux: "
JXL_AUX_STREAMS=$aux python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-gather --stages $st 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['config'].get('single_frame_ms'))"
done; done

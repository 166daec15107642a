#!/bin/bash
# the round's bench figures, each the verbatim JSON line of the named command (GPU box): bash tools/bench_set.sh r3
R=${1:-r5}
O=gpurun_out/bench_$R
mkdir -p $O
L=tests/golden/samples_large
run() { python bench.py "${@:2}" > $O/$1.json 2> $O/$1.err; python -c "
import json,sys
d=json.loads(open('$O/$1.json').read().strip().splitlines()[-1]); print('%-28s value %9.1f  ms/step %.4f' % ('$1', d['value'], d['ms_per_step']))"; }
run ${R}_bench_vardct4k
run ${R}_bench_vardct4k_single --frames-per-gpu 1 --no-cpu-baseline --no-end-to-end --no-also
run ${R}_bench_vardct4k_epf3 --epf-iters 3 --no-cpu-baseline --no-end-to-end --no-also
run ${R}_bench_vardct4k_epf1 --epf-iters 1 --no-cpu-baseline --no-end-to-end --no-also
run ${R}_bench_vardct8k_pq --workload vardct8k_pq --frames-per-gpu 2
run ${R}_bench_modular1080p --workload modular1080p
run ${R}_bench_modular8k --workload modular8k
run ${R}_bench_modular1080p_single --workload modular1080p --frames-per-gpu 1
run ${R}_bench_modular8k_single --workload modular8k --frames-per-gpu 1
run ${R}_bench_real_sollevante4k --workload jxlfile --input $L/sollevante-hdr.jxl --no-cpu-baseline
run ${R}_bench_real_bbb720p_batch --workload jxlfile --input tests/golden/samples/bbb.jxl --batch --no-cpu-baseline
run ${R}_bench_real_lenna512_batch --workload jxlfile --input tests/golden/samples/lenna.jxl --batch --no-cpu-baseline

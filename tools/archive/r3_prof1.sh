#!/bin/bash
# kernel timeline of a single 4K frame (steady state), optional env passes through
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/tl && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o p -- python3 $R/bench.py --steps 6 --warmup 2 --frames-per-gpu 1 --no-cpu-baseline --no-end-to-end --no-gather > /tmp/tl.log 2>&1
python3 $R/tools/kernel_timeline.py /tmp/tl ${1:-14}

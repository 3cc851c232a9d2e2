#!/bin/bash
# one-query kernel timeline of the current (or SPIRAL_LIB) build: bash tools/trace_query.sh <outname> [query index]
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/trace; mkdir -p $O
rm -rf $O/kt
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-config3 --lanes 1 > $O/$1.log 2>&1
python tools/trace_summary.py $O/kt/*/*_kernel_trace.csv --timeline --query ${2:-11} > $O/$1_timeline.txt
python tools/kernel_avg.py $O/kt/*/*_kernel_trace.csv "" --by-grid > $O/$1_kernel_avg.txt
rm -rf $O/kt
tail -1 $O/$1.log | cut -c1-400

#!/bin/bash
# kernel timeline of one batch of B whole queries (run_query_batch): bash tools/batch_timeline.sh B <outfile>
cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/trace; mkdir -p $O
rm -rf $O/ktb
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/ktb -- python3 tools/batch_query.py $1 --reps=10 > $O/ktb.log 2>&1
python tools/trace_summary.py $O/ktb/*/*_kernel_trace.csv --timeline --query 20 > $2
rm -rf $O/ktb

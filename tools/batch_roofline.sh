#!/bin/bash
# One B-query batch (run_query_batch at config 2): kernel-trace timeline + FETCH_SIZE / WRITE_SIZE in their own passes -> per-kernel roofline JSON.
# usage (through gpurun, from the repo root): bash tools/batch_roofline.sh r06 [B=8]
set -u
R=${1:-rXX}; B=${2:-8}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/batch; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/batch_query.py $B --reps=10 > $O/kt.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pf -- python3 tools/batch_query.py $B --reps=2 > $O/pf.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pw -- python3 tools/batch_query.py $B --reps=2 > $O/pw.log 2>&1
python tools/trace_summary.py $O/kt/*/*_kernel_trace.csv --timeline --query 20 > $O/${R}_one_query_timeline_B$B.txt
python tools/batch_bytes.py $O/kt/*/*_kernel_trace.csv $O/pf/*/*_counter_collection.csv $O/pw/*/*_counter_collection.csv --lanes=$B > $O/${R}_batch_kernel_bytes_B$B.json 2> $O/bb.err
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_ANY --kernel-trace --output-format csv -d $O/ps -- python3 tools/batch_query.py $B --reps=2 > $O/ps.log 2>&1
python tools/pmc_summary.py $O/ps/*/*_counter_collection.csv > $O/${R}_sq_counters_batch$B.json
rm -rf $O/kt $O/pf $O/pw $O/ps
tail -3 $O/kt.log; cat $O/bb.err; head -c 3000 $O/${R}_batch_kernel_bytes_B$B.json

cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/b4; mkdir -p $O
python tools/batch_query.py 1 2 3 4 > $O/batch_times.txt 2>&1
for B in ${BS:-4}; do
timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt$B -- python3 tools/batch_query.py $B --reps=10 > $O/rocprof$B.log 2>&1
python tools/trace_summary.py $O/kt$B/*/*_kernel_trace.csv --timeline --query 20 > $O/timeline_B$B.txt
rm -rf $O/kt$B
done
cat $O/batch_times.txt; cat $O/timeline_B4.txt

cd /tmp && export TMPDIR=/tmp; cd "$GRAFT_REPO_ROOT"
O=gpurun_out/evidence2; mkdir -p $O
python bench.py > $O/r06_bench.json 2> $O/r06_bench.err
for B in 4 8; do bash tools/batch_roofline.sh r06 $B > $O/batch_roofline_B$B.log 2>&1; cp gpurun_out/batch/r06_one_query_timeline_B$B.txt gpurun_out/batch/r06_batch_kernel_bytes_B$B.json gpurun_out/batch/r06_sq_counters_batch$B.json $O/; done
python tools/batch_query.py 1 2 3 4 5 6 7 8 > $O/r06_batch_times.txt 2>&1
python tools/batch_query.py 2 4 8 --streams=1 > $O/r06_batch_times_own_streams.txt 2>&1
python tools/batch_overlap.py --groups=2 --lanes=8 --reps=30 2>&1 | tail -2 > $O/r06_two_batches_in_flight.txt
python tools/sweep_in_situ_batch.py 8 > $O/r06_sweep_in_situ_batch.txt 2>&1
python bench.py --workload config3 --steps 10 --no-cpu-baseline > $O/r06_bench_config3.json 2>/dev/null
python bench.py --workload stream-instance --steps 10 --no-cpu-baseline > $O/r06_bench_stream_one_instance.json 2>/dev/null
cat $O/r06_batch_times.txt $O/r06_two_batches_in_flight.txt | grep -v amdgpu; head -14 $O/r06_one_query_timeline_B8.txt

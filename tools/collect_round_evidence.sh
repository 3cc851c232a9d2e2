#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/evidence (copy what is to be judged into profiles/).
# usage (from the repo root, through gpurun): bash tools/collect_round_evidence.sh r06
set -u
R=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/evidence; mkdir -p $O
F="--no-cpu-baseline --no-config3 --lanes 1"
python bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 3 $F > $O/${R}_bench_under_rocprof.log 2>&1
cp $O/kt/*/*_kernel_stats.csv $O/${R}_bench_kernel_stats.csv
# the trace holds, in order: 2 priming queries, the W + K steps of the no-pre-warm pass (3 + 20), --prewarm (40) queries, 3 warm-up steps, 20 timed steps, then the
# standalone sweep launches.  Query 40 is a pre-warm query = a whole-query graph replay (no gaps); query 73 is timed step 5, a SAMPLED step, whose hipEventRecords
# between the stages show up as gaps
python tools/trace_summary.py $O/kt/*/*_kernel_trace.csv --timeline --query 40 > $O/${R}_one_query_timeline.txt
python tools/trace_summary.py $O/kt/*/*_kernel_trace.csv --timeline --query 73 > $O/${R}_one_query_timeline_sampled_step.txt
python tools/kernel_avg.py $O/kt/*/*_kernel_trace.csv "" --by-grid > $O/${R}_kernel_avg_by_grid.txt
rm -rf $O/kt
# B whole queries per launch sequence (run_query_batch): times for B = 1 .. 8, the same with every lane on its own stream, and per batch size the timeline of one batch +
# its per-kernel roofline (kernel-trace durations; FETCH_SIZE and WRITE_SIZE each in its own pass; tools/batch_roofline.sh -> gpurun_out/batch)
python tools/batch_query.py 1 2 3 4 5 6 7 8 > $O/${R}_batch_times.txt 2>&1
python tools/batch_query.py 2 4 8 --streams=1 > $O/${R}_batch_times_own_streams.txt 2>&1
for B in 4 8; do
bash tools/batch_roofline.sh $R $B > $O/batch_roofline_B$B.log 2>&1
cp gpurun_out/batch/${R}_one_query_timeline_B$B.txt gpurun_out/batch/${R}_batch_kernel_bytes_B$B.json gpurun_out/batch/${R}_sq_counters_batch$B.json $O/
done
python tools/sweep_in_situ_batch.py 8 > $O/${R}_sweep_in_situ_batch.txt 2>&1
# the batched sweep on the matrix cores: bit-identity against the single-query sweep on random inputs + kernel durations for 1 .. 8 queries per pass
# (rocprofv3 kernel trace: the tool's own wall figures include the host's event ordering of eight streams), then its counters
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/ktm -- python3 tools/sweep_mfma_check.py --reps=4 > $O/${R}_sweep_mfma_check.txt 2>&1
python3 - >> $O/${R}_sweep_mfma_check.txt <<PY
import csv, glob
for r in csv.DictReader(open(glob.glob("$O/ktm/*/*_kernel_stats.csv")[0])):
    if "sweep" in r["Name"]: print("%-50s calls %4s avg %8.1f us  min %8.1f us" % (r["Name"].split("(")[0][-50:], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3))
PY
rm -rf $O/ktm
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pm1 -- python3 tools/sweep_mfma_check.py --reps=2 > $O/pm1.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pm2 -- python3 tools/sweep_mfma_check.py --reps=2 > $O/pm2.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pm3 -- python3 tools/sweep_mfma_check.py --reps=2 > $O/pm3.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pm4 -- python3 tools/sweep_mfma_check.py --reps=2 > $O/pm4.log 2>&1
python tools/pmc_summary.py $O/pm1/*/*_counter_collection.csv $O/pm2/*/*_counter_collection.csv $O/pm3/*/*_counter_collection.csv $O/pm4/*/*_counter_collection.csv > $O/${R}_sweep_mfma_counters.json
rm -rf $O/pm1 $O/pm2 $O/pm3 $O/pm4
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 4 --warmup 1 $F --no-graphs > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 4 --warmup 1 $F --no-graphs > $O/pmc_write.log 2>&1
cp $O/pmc_fetch/*/*_counter_collection.csv $O/${R}_pmc_fetch_size_counter_collection.csv
cp $O/pmc_write/*/*_counter_collection.csv $O/${R}_pmc_write_size_counter_collection.csv
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sq1 -- python3 bench.py --steps 2 --warmup 1 $F --no-graphs > $O/pmc_sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -- python3 bench.py --steps 2 --warmup 1 $F --no-graphs > $O/pmc_sq2.log 2>&1
python tools/pmc_summary.py $O/pmc_sq1/*/*_counter_collection.csv $O/pmc_sq2/*/*_counter_collection.csv > $O/${R}_sq_counters_per_kernel.json
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2
# the fold's forms that remain, alternating on this box (group times of hipGraph replays, tools/stage_ab.py)
python tools/stage_ab.py "" "fold_pair=0" "fold_chain=0" "" "fold_pair=0" > $O/${R}_fold_forms_ab.txt 2>&1
( python tools/digits_time.py; python tools/digits_time.py fwd2=1; python tools/digits_time.py fwd2=0 ) > $O/${R}_digits_time_fwd2.txt 2>&1
python tools/cpu_oracle_scaling.py 1 8 16 32 > $O/${R}_cpu_oracle_scaling.txt 2>&1
python tools/shard_estimate.py > $O/${R}_shard_estimate.txt 2>&1
python bench.py --workload config3 --steps 10 --no-cpu-baseline > $O/${R}_bench_config3.json 2>/dev/null
python bench.py --workload stream --steps 5 --warmup 1 > $O/${R}_bench_stream.json 2>/dev/null
python bench.py --workload stream-instance --steps 10 --no-cpu-baseline > $O/${R}_bench_stream_one_instance.json 2>/dev/null
python bench.py --workload stream --gpus 2 --backend gloo --shared-device --steps 3 --warmup 1 --nu1 6 --nu2 6 > $O/${R}_bench_stream_selflaunch_2ranks_1gpu.json 2>/dev/null
( tools/mac_gather_probe 8 2 4 8 16 32 64 128; tools/mac_gather_probe 24 6 12 24 48 ) > $O/${R}_mac_gather_probe.txt 2>&1
python bench.py --workload pack --steps 10 --warmup 2 > $O/${R}_bench_pack.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --shared-device --steps 10 > $O/${R}_bench_selflaunch_2ranks_1gpu.json 2>/dev/null
python bench.py --gpus 8 --backend gloo --shared-device --steps 5 --no-config3 --no-replicas > $O/${R}_bench_selflaunch_8ranks_1gpu.json 2>/dev/null
SPIRAL_BENCH_INJECT_HANG=pg-setup:1 SPIRAL_BENCH_PG_WATCHDOG_S=20 python bench.py --gpus 2 --backend gloo --shared-device --steps 3 --no-config3 --no-replicas > $O/${R}_bench_injected_pg_hang_2ranks.json 2> $O/${R}_bench_injected_pg_hang_2ranks.err; echo "exit code $?" >> $O/${R}_bench_injected_pg_hang_2ranks.json
SPIRAL_BENCH_INJECT_HANG=pipelined SPIRAL_BENCH_WATCHDOG_S=25 SPIRAL_BENCH_PARTIAL=$O/${R}_bench_partial_file.json python bench.py --gpus 2 --backend gloo --shared-device --steps 5 --no-config3 --no-replicas > $O/${R}_bench_injected_hang_2ranks.json 2> $O/${R}_bench_injected_hang_2ranks.err; echo "exit code $?" >> $O/${R}_bench_injected_hang_2ranks.json
ls -la $O | head -60

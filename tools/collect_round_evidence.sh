#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/evidence (copy what is to be judged into profiles/).
# usage (from the repo root, through gpurun): bash tools/collect_round_evidence.sh r02
set -u
R=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/evidence; mkdir -p $O
python bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --lanes 1 > $O/${R}_bench_under_rocprof.log 2>&1
cp $O/kt/*/*_kernel_stats.csv $O/${R}_bench_kernel_stats.csv
python tools/trace_summary.py $O/kt/*/*_kernel_trace.csv --timeline --query 10 > $O/${R}_one_query_timeline.txt
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-graphs --lanes 1 > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-graphs --lanes 1 > $O/pmc_write.log 2>&1
cp $O/pmc_fetch/*/*_counter_collection.csv $O/${R}_pmc_fetch_size_counter_collection.csv
cp $O/pmc_write/*/*_counter_collection.csv $O/${R}_pmc_write_size_counter_collection.csv
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sq1 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graphs --lanes 1 > $O/pmc_sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-graphs --lanes 1 > $O/pmc_sq2.log 2>&1
python tools/pmc_summary.py $O/pmc_sq1/*/*_counter_collection.csv $O/pmc_sq2/*/*_counter_collection.csv > $O/${R}_sq_counters_per_kernel.json
tools/ubench_valu > $O/${R}_ubench_valu.txt 2>&1
tools/occupancy_probe > $O/${R}_occupancy_probe.txt 2>&1
python tools/cpu_oracle_scaling.py 1 8 16 32 64 128 > $O/${R}_cpu_oracle_scaling.txt 2>&1
python tools/shard_estimate.py > $O/${R}_shard_estimate.txt 2>&1
python bench.py --workload config3 --steps 10 --no-cpu-baseline > $O/${R}_bench_config3.json 2>/dev/null
python bench.py --workload stream --steps 10 --no-cpu-baseline > $O/${R}_bench_stream.json 2>/dev/null
python bench.py --workload pack --steps 10 --warmup 2 > $O/${R}_bench_pack.json 2>/dev/null
tools/ntt_valu_probe > $O/${R}_ntt_valu_probe.txt 2>&1
tools/mem_bw_probe > $O/${R}_mem_bw_probe.txt 2>&1
ls -la $O | head -40

#!/bin/bash
# Collects the round's evidence on the GPU box into gpurun_out/evidence (copy what is to be judged into profiles/).
# usage (from the repo root, through gpurun): bash tools/collect_round_evidence.sh r03
# (tools/variants/libspiral_prio1.so / prio3.so: build first with  tools/build_variants.sh ntt.hip prio1=-DNTT_PRIO=1 prio3=-DNTT_PRIO=3)
set -u
R=${1:-rXX}
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/evidence; mkdir -p $O
F="--no-cpu-baseline --no-config3 --lanes 1"
python bench.py > $O/${R}_bench.json 2> $O/${R}_bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --steps 20 --warmup 3 $F > $O/${R}_bench_under_rocprof.log 2>&1
cp $O/kt/*/*_kernel_stats.csv $O/${R}_bench_kernel_stats.csv
python tools/trace_summary.py $O/kt/*/*_kernel_trace.csv --timeline --query 10 > $O/${R}_one_query_timeline.txt
rm -rf $O/kt
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -- python3 bench.py --steps 4 --warmup 1 $F --no-graphs > $O/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -- python3 bench.py --steps 4 --warmup 1 $F --no-graphs > $O/pmc_write.log 2>&1
cp $O/pmc_fetch/*/*_counter_collection.csv $O/${R}_pmc_fetch_size_counter_collection.csv
cp $O/pmc_write/*/*_counter_collection.csv $O/${R}_pmc_write_size_counter_collection.csv
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $O/pmc_sq1 -- python3 bench.py --steps 2 --warmup 1 $F --no-graphs > $O/pmc_sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_sq2 -- python3 bench.py --steps 2 --warmup 1 $F --no-graphs > $O/pmc_sq2.log 2>&1
python tools/pmc_summary.py $O/pmc_sq1/*/*_counter_collection.csv $O/pmc_sq2/*/*_counter_collection.csv > $O/${R}_sq_counters_per_kernel.json
rm -rf $O/pmc_fetch $O/pmc_write $O/pmc_sq1 $O/pmc_sq2
timeout 120 tools/grid_sync_probe > $O/${R}_grid_sync_probe.txt 2>&1
timeout 120 tools/mac_gather_probe > $O/${R}_mac_gather_probe.txt 2>&1
python tools/cpu_oracle_scaling.py 1 8 16 32 > $O/${R}_cpu_oracle_scaling.txt 2>&1
python tools/shard_estimate.py > $O/${R}_shard_estimate.txt 2>&1
python bench.py --workload config3 --steps 10 --no-cpu-baseline > $O/${R}_bench_config3.json 2>/dev/null
python bench.py --workload stream --steps 10 --no-cpu-baseline > $O/${R}_bench_stream.json 2>/dev/null
python bench.py --workload pack --steps 10 --warmup 2 > $O/${R}_bench_pack.json 2>/dev/null
python bench.py --gpus 2 --backend gloo --shared-device --steps 10 > $O/${R}_bench_selflaunch_2ranks_1gpu.json 2>/dev/null
# issue-priority A/B (NTT_PRIO hook, ntt_device.h), alternating with the default build on this box
for rep in 1 2; do
  python bench.py $F > $O/prio_base_$rep.json 2>/dev/null
  for v in prio1 prio3; do [ -f tools/variants/libspiral_$v.so ] && SPIRAL_LIB=tools/variants/libspiral_$v.so python tools/variant_bench.py $F > $O/prio_${v}_$rep.json 2>/dev/null; done
done
python - > $O/${R}_ntt_prio_ab.txt <<PY
import json, glob
print("# NTT_PRIO A/B (s_setprio raised from a pass's LDS stores to the next pass's twiddle loads, 0 for the butterflies; -DNTT_PRIO=1|3 builds of ntt.hip),")
print("# config 2, one box, alternating: python bench.py --no-cpu-baseline --no-config3 --lanes 1  /  SPIRAL_LIB=tools/variants/libspiral_prioN.so python tools/variant_bench.py <same>")
for f in sorted(glob.glob("$O/prio_*.json")):
    try:
        d = json.loads(open(f).read().strip().splitlines()[-1]); n = d.get("roofline_ntt") or {}
        print(f"{f.split('/')[-1]:22s} {d['value']:.4f} ms/query  stages_us {d['stages_us']}  to_ntt {n.get('forward_to_ntt')} from_ntt {n.get('inverse_from_ntt')} digits {n.get('forward_digits')} ns")
    except Exception as e:
        print(f, "ERR", e)
PY
rm -f $O/prio_*.json
ls -la $O | head -50

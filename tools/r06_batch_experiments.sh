#!/bin/bash
# round 6, batch kernels: where the conversion products and the in-situ matrix-core sweep lose their time (variants built by tools/build_variants.sh)
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/exp; mkdir -p $O
V=tools/variants
trace() {  # name, env...: kernel-trace of 10 eight-query batches, the durations of the kernels of interest in batch 20 of the trace
  local name=$1; shift
  env "$@" python3 -c 'print("ok")' > /dev/null
  ( export "$@"; timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt_$name -- python3 tools/batch_query.py 8 --reps=10 > $O/kt_$name.log 2>&1 )
  python tools/trace_summary.py $O/kt_$name/*/*_kernel_trace.csv --query 20 | head -9 | sed "s/^/[$name] /"
  tail -1 $O/kt_$name.log | sed "s/^/[$name] /"
  rm -rf $O/kt_$name
}
trace product SPIRAL_X=0
trace tuning_default SPIRAL_LIB=$V/libspiral_tuning.so
trace s2m_wide_tile SPIRAL_LIB=$V/libspiral_tuning.so SPIRAL_S2M_WIDE_MIN=4
trace gsw_first SPIRAL_LIB=$V/libspiral_gswfirst.so
trace quad_ablation SPIRAL_LIB=$V/libspiral_quadabl.so
python tools/sweep_in_situ_batch.py 8 2>&1 | sed "s/^/[in-situ product] /"
SPIRAL_LIB=$V/libspiral_quadabl.so python tools/sweep_in_situ_batch.py 8 2>&1 | sed "s/^/[in-situ quad ablation] /"

#!/bin/bash
set -u
cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/exp2; mkdir -p $O
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "chain_schedules or run_query_batch or instances or stage" 2>&1 | tail -3
python -m pytest tests/test_gpu_dist2.py -m gpu -x -q -k "never_joined or headline or stream_item" 2>&1 | tail -5
for t in 8192 4096 3072 2048; do echo "fwd2_min=$t"; python tools/batch_query.py 1 4 8 --opt:fwd2_min=$t 2>&1 | grep "B="; done
echo "--- consumer-side gather, K = 8 (even expansion rounds 0..5: 2 rows x 2^r ciphertexts)"; tools/mac_gather_probe 8 2 4 8 16 32 64 128
echo "--- K = 24 (pair-form fold, narrow rounds: 6 np outputs)"; tools/mac_gather_probe 24 6 12 24 48
python bench.py --workload stream --steps 5 --warmup 1 > $O/r06_bench_stream.json 2> $O/bench_stream.err; tail -c 2500 $O/r06_bench_stream.json; tail -3 $O/bench_stream.err

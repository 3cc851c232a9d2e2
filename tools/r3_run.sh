# scratch driver for one gpurun call (not part of the product): bash tools/r3_run.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r3}; O=gpurun_out/$T; mkdir -p $O
timeout 300 python bench.py --gpus 2 --shared-device --backend nccl --steps 4 --warmup 1 --no-config3 > $O/bench_nccl2_shared.json 2> $O/bench_nccl2_shared.err; echo "rc=$?"
tail -c 1500 $O/bench_nccl2_shared.err; head -c 600 $O/bench_nccl2_shared.json
rocm-smi --showtopo 2>/dev/null | head -20; rocminfo | grep -c "gfx950"

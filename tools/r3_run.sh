# scratch driver for one gpurun call (not part of the product): bash tools/r3_run.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r3}; O=gpurun_out/$T; mkdir -p $O
( time SPIRAL_FUZZ_SETS=400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_pack.py -k "random" -q -m gpu ) > $O/fuzz_soak.log 2>&1
tail -6 $O/fuzz_soak.log
( time python -m pytest tests -m gpu -q ) > $O/pytest_full.log 2>&1
tail -16 $O/pytest_full.log

# scratch driver for one gpurun call (not part of the product): bash tools/r3_run.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r3}; O=gpurun_out/$T; mkdir -p $O
python -m pytest tests -m "gpu and not slow" -x -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -22 $O/pytest.log

cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r4}; O=gpurun_out/$T; mkdir -p $O
python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $O/smoke.log
python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -4 $O/pytest.log
( time python bench.py ) > $O/bench_default.json 2> $O/bench_default.err
tail -3 $O/bench_default.err
python - <<PY
import json
d=json.loads(open('$O/bench_default.json').read().strip().splitlines()[-1])
print(d['value'], d['stages_us'], d['roofline']['frac'], d['pipelined']['queries_per_s'], d['pipelined']['batched_sweep']['2'], d['also']['config3']['value'], d['cpu_baseline']['value'], d['cpu_baseline']['all_cores']['value'])
PY

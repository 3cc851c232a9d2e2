# scratch driver for one gpurun call (not part of the product): bash tools/r3_run.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r3}; O=gpurun_out/$T; mkdir -p $O
F="--no-cpu-baseline --no-config3 --lanes 1"
python bench.py $F > $O/bench_base_1.json 2>> $O/bench.err
python bench.py $F --event-every 1 > $O/bench_ev1.json 2>> $O/bench.err
python bench.py $F --overlap 1 > $O/bench_overlap_lowprio.json 2>> $O/bench.err
SPIRAL_SIDE_PRIO_DEFAULT=1 python bench.py $F --overlap 1 > $O/bench_overlap_defprio.json 2>> $O/bench.err
python bench.py $F > $O/bench_base_2.json 2>> $O/bench.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['stages_us'])
    except Exception as e: print(f, 'ERR', e)
PY

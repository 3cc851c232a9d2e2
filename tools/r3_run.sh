# scratch driver for one gpurun call (not part of the product): bash tools/r3_run.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r3}; O=gpurun_out/$T; mkdir -p $O
python -m pytest tests -m "gpu and not slow" -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
F="--no-cpu-baseline --no-config3 --lanes 1"
python bench.py $F > $O/bench_new_1.json 2>> $O/bench.err
( time python -m spiral_amd.scheme --fit ) > $O/fit.log 2>&1
( time python -m spiral_amd.scheme --fit-pack ) > $O/fit_pack.log 2>&1
cp spiral_amd/cost_model_mi355x.json $O/cost_model_mi355x.json
python -m spiral_amd.scheme --select 20,256 --optimize-for tput --run --trials 3 --analyze-deviation > $O/select_deviation_20_256.txt 2>&1
python -m spiral_amd.scheme --select 18,30000 --variant spiral-pack --optimize-for tput --one-gpu --run --trials 2 > $O/select_18_30000_spiral-pack.json 2>$O/select_pack.err
python bench.py $F > $O/bench_new_2.json 2>> $O/bench.err
tail -3 $O/pytest.log; tail -5 $O/fit.log; tail -5 $O/fit_pack.log; tail -25 $O/select_deviation_20_256.txt
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['stages_us'])
    except Exception as e: print(f, 'ERR', e)
PY

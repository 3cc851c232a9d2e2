# scratch driver for one gpurun call (not part of the product): bash tools/r3_run.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r3}; O=gpurun_out/$T; mkdir -p $O
F="--no-cpu-baseline --no-config3 --lanes 1"
for rep in 1 2; do
python bench.py $F > $O/bench_base_$rep.json 2>> $O/bench.err
for v in macu4 macu7; do SPIRAL_LIB=tools/variants/libspiral_$v.so python tools/variant_bench.py $F > $O/bench_${v}_$rep.json 2>> $O/bench.err; done
done
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['stages_us'])
    except Exception as e: print(f, 'ERR', e)
PY

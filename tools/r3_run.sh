# scratch driver for one gpurun call (not part of the product): bash tools/r3_run.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r3}; O=gpurun_out/$T; mkdir -p $O
F="--no-cpu-baseline --no-config3 --lanes 1"
for rep in 1 2 3; do
python bench.py $F > $O/bench_new_$rep.json 2>> $O/bench.err
SPIRAL_LIB=tools/variants/libspiral_nobin8.so python tools/variant_bench.py $F > $O/bench_nobin8_$rep.json 2>> $O/bench.err
done
python bench.py --workload stream --steps 10 $F > $O/bench_stream_new.json 2>> $O/bench.err
SPIRAL_LIB=tools/variants/libspiral_nobin8.so python tools/variant_bench.py --workload stream --steps 10 $F > $O/bench_stream_nobin8.json 2>> $O/bench.err
python - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['stages_us'], d.get('reference_buckets_us_eager'))
    except Exception as e: print(f, 'ERR', e)
PY

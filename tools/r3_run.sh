# scratch driver for one gpurun call (not part of the product): bash tools/r3_run.sh <tag>
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
T=${1:-r3}; O=gpurun_out/$T; mkdir -p $O
python -m pytest tests -m "gpu and not slow" -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
F="--no-cpu-baseline --no-config3 --lanes 1"
for rep in 1 2; do
python bench.py $F > $O/bench_new_$rep.json 2>> $O/bench_new.err
for v in r2final chain5 chain6; do SPIRAL_LIB=tools/variants/libspiral_$v.so python tools/variant_bench.py $F > $O/bench_${v}_$rep.json 2> $O/bench_$v.err; done
done
echo "nproc $(nproc)"; cat /sys/fs/cgroup/cpu.max 2>/dev/null; taskset -p $$; grep -c processor /proc/cpuinfo; cat /proc/loadavg
tail -3 $O/pytest.log; python - <<PY
import json,glob
for f in sorted(glob.glob('$O/bench_*.json')):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d['value'], d['stages_us'], (d.get('roofline_ntt') or {}).get('inverse_from_ntt'), (d.get('roofline_ntt') or {}).get('forward_digits'))
    except Exception as e: print(f, 'ERR', e)
PY

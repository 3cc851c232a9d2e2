#!/bin/bash
# per-kernel totals of one bench workload: bash tools/profile_workload.sh <workload> [extra bench flags]   (run from the repo root on the GPU box)
export TMPDIR=/tmp
W=${1:-config2}; shift
O=gpurun_out/prof_$W; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python3 bench.py --workload $W --steps 8 --warmup 2 --no-cpu-baseline --lanes 1 --event-every 1 "$@" > $O/bench.log 2>&1
python3 - "$O" <<'PY'
import csv, glob, sys
f = glob.glob(sys.argv[1] + "/kt/*/*_kernel_stats.csv")[0]
rows = list(csv.DictReader(open(f)))
for r in rows[:14]:
    print(f"{float(r['TotalDurationNs'])/1e3:12.1f} us total {int(r['Calls']):6d} calls {float(r['AverageNs'])/1e3:10.1f} us avg  {r['Name'][:100]}")
PY
tail -c 400 $O/bench.log

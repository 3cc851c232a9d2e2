# kernel durations of the batched sweeps (rocprofv3 kernel trace): bash tools/mfma_ktime.sh [SPIRAL_LIB path]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/mfma_kt
rm -rf $O; mkdir -p $O
cd $R
[ -n "$1" ] && export SPIRAL_LIB=$1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 tools/sweep_mfma_check.py --reps=4 > $O/log 2>&1
python3 - <<PY
import csv,glob,collections
f=glob.glob("$O/p/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "sweep" in r["Name"]: print("%-60s calls %4s avg %8.1f us  min %8.1f" % (r["Name"][:60], r["Calls"], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
rm -rf $O/p

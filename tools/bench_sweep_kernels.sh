cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/bk
rm -rf $O; mkdir -p $O
cd $R
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/p -- python3 bench.py --steps 5 --warmup 2 > $O/log 2>$O/err
python3 - <<PY
import csv,glob
f=glob.glob("$O/p/*/*_kernel_stats.csv")[0]
for r in csv.DictReader(open(f)):
    if "sweep" in r["Name"] or "limb" in r["Name"]: print("%-70s calls %4s avg %8.1f us" % (r["Name"][:70], r["Calls"], float(r["AverageNs"])/1e3))
PY
rm -rf $O/p

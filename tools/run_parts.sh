cd /tmp && export TMPDIR=/tmp
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/parts; mkdir -p $O
python tools/batch_query.py 4 > /dev/null 2>&1
for P in 4 5 6 7; do
SPIRAL_BATCH_PARTS=$P python tools/batch_query.py 1 2 4 > $O/times_parts$P.txt 2>&1
SPIRAL_BATCH_PARTS=$P timeout 300 rocprofv3 --kernel-trace --output-format csv -d $O/kt -- python3 tools/batch_query.py 4 --reps=10 > $O/rocprof.log 2>&1
python - $O/kt/*/*_kernel_trace.csv > $O/tail_parts$P.txt <<'PY'
import csv,sys
rows=sorted(csv.DictReader(open(sys.argv[1])),key=lambda r:int(r["Start_Timestamp"]))
ends=[i for i,r in enumerate(rows) if "rescale" in r["Kernel_Name"]]
i=ends[len(ends)//2]
for r in rows[i-14:i+1]:
    n=r["Kernel_Name"].replace("spiral::","").replace("void ","").split("(")[0][:40]
    print(f"dur={(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:7.1f} grid={r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:>4s}x{r['Grid_Size_Z']:>3s} {n}")
PY
rm -rf $O/kt
echo "parts=$P"; cat $O/times_parts$P.txt; cat $O/tail_parts$P.txt
done

#!/usr/bin/env python3
"""Summarise a rocprofv3 --kernel-trace CSV: per-kernel totals and the timeline of one query.
usage: tools/trace_summary.py gpurun_out/<dir>/<name>_kernel_trace.csv [--timeline] [--query K]
The default query is the middle one of the trace: a steady-state step of bench.py's timed loop.  (The last two queries
of a bench.py trace are the eager detail pass, whose hipEventRecords between stages show up as ~6 us gaps.)"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sweeps = [i for i, r in enumerate(rows) if "sweep_kernel" in r["Kernel_Name"] or "sweep_mfma_kernel" in r["Kernel_Name"]]
nq = len(sweeps)
which = int(sys.argv[sys.argv.index("--query") + 1]) if "--query" in sys.argv else nq // 2
last = sweeps[which]
# a query = the launches after the previous response switch up to and including this one's
ends = [i for i, r in enumerate(rows) if "rescale" in r["Kernel_Name"]]
start = max([i for i in ends if i < last], default=-1) + 1
end = min([i for i in ends if i > last], default=len(rows) - 1)
q = rows[start : end + 1]
t0 = int(q[0]["Start_Timestamp"])
tot = defaultdict(lambda: [0, 0.0])
busy = 0.0
for r in q:
    d = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    n = r["Kernel_Name"].replace("spiral::", "").replace("void ", "").split("(")[0]
    tot[n][0] += 1
    tot[n][1] += d
    busy += d
wall = (int(q[-1]["End_Timestamp"]) - t0) / 1e3
print(f"queries in trace: {nq}; query {which}: {len(q)} launches, wall {wall:.1f} us, kernel-busy {busy:.1f} us")
for n, (c, d) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"  {d:8.1f} us  {c:4d} x  {n}")
if "--timeline" in sys.argv:
    pe = t0
    for r in q:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        n = r["Kernel_Name"].replace("spiral::", "").replace("void ", "").split("(")[0][:44]
        print(f"{(s - t0) / 1e3:9.1f} gap={(s - pe) / 1e3:6.1f} dur={(e - s) / 1e3:7.1f} grid={r['Grid_Size_X']:>8s}x{r['Grid_Size_Y']:>4s}x{r['Grid_Size_Z']:>3s} {n}")
        pe = e

#!/usr/bin/env python3
"""per-kernel sums of rocprofv3 --pmc counters: tools/pmc_summary.py <counter_collection.csv> [...] -> JSON on stdout
(kernels keyed by short name and grid size; counters summed over the dispatches of that key, `dispatches` says how many)"""
import csv
import json
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(float))
n = defaultdict(set)
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"].replace("spiral::", "").replace("void ", "").split("(")[0]
        key = f"{name} grid={r['Grid_Size']}"
        acc[key][r["Counter_Name"]] += float(r["Counter_Value"])
        n[key].add(r["Dispatch_Id"])
out = {}
for k in sorted(acc):
    d = {c: v for c, v in sorted(acc[k].items())}
    d["dispatches"] = len(n[k])
    out[k] = d
print(json.dumps(out, indent=1))

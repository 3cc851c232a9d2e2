#!/usr/bin/env python3
"""average duration per kernel name (optionally per grid size) in a rocprofv3 --kernel-trace CSV
usage: tools/kernel_avg.py trace.csv [name-substring] [--by-grid]"""
import csv
import sys
from collections import defaultdict

rows = list(csv.DictReader(open(sys.argv[1])))
pat = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else ""
by_grid = "--by-grid" in sys.argv
acc = defaultdict(list)
for r in rows:
    n = r["Kernel_Name"].replace("spiral::", "").replace("void ", "").split("(")[0]
    if pat not in n:
        continue
    key = (n, int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])) if by_grid else (n, 0)
    acc[key].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for (n, g), v in sorted(acc.items()):
    v.sort()
    print(f"{n[:40]:40s} grid={g:9d} n={len(v):5d} median={v[len(v)//2]:8.2f} us  min={v[0]:8.2f}")

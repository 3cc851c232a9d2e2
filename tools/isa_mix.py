#!/usr/bin/env python3
"""static instruction mix of one kernel in a hipcc -S listing: tools/isa_mix.py file.s <mangled-name-substring> [--top N]
(straight-line kernels: the static count is the per-thread dynamic count; loops are counted once)"""
import collections
import re
import sys

src = open(sys.argv[1]).read().split("\n")
pat = sys.argv[2]
top = int(sys.argv[sys.argv.index("--top") + 1]) if "--top" in sys.argv else 40
start = next(i for i, l in enumerate(src) if re.match(r"^_Z\S*:", l) and pat in l)
end = next(i for i in range(start, len(src)) if src[i].strip().startswith("s_endpgm"))
ops = collections.Counter()
for l in src[start + 1:end + 1]:
    l = l.split(";")[0].strip()
    if not l or l.startswith(".") or l.endswith(":"):
        continue
    ops[l.split()[0]] += 1
total = sum(ops.values())
cls = collections.Counter()
for o, n in ops.items():
    c = ("valu" if o.startswith("v_") else "salu" if o.startswith("s_") else "lds" if o.startswith("ds_") else
         "vmem" if o.startswith(("global_", "buffer_", "flat_", "scratch_")) else "other")
    cls[c] += n
print(src[start][:90], "instructions:", total, dict(cls))
for o, n in ops.most_common(top):
    print(f"  {n:5d} {o}")

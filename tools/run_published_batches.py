#!/usr/bin/env python3
"""every published BASE parameter set (query compression or direct upload, no response packing) through ./spiral --batch 8: eight clients with their own keys and
indices answered by one launch sequence (run_query_batch; the sweep one pass on the matrix cores where the geometry allows), each decoded and checked.
usage: python tools/run_published_batches.py [--batch 8] [--seed N]"""
import os, re, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import scheme

seed = sys.argv[sys.argv.index("--seed") + 1] if "--seed" in sys.argv else "3"
batch = sys.argv[sys.argv.index("--batch") + 1] if "--batch" in sys.argv else "8"
print(f"{'set':28s} {'nu1':>3s} {'nu2':>3s} {'matrix cores':>12s} {'correct':>16s} {'batch_us':>9s} {'queries/s':>9s} {'1-query sweep us':>16s}")
for work, variants in scheme.PUBLISHED.items():
    for variant, params in variants.items():
        if scheme.is_high_rate(params):
            continue
        argv, env = scheme.command(params, 12345 % (1 << (params["nu_1"] + params["nu_2"])), True, seed)
        r = subprocess.run(argv + ["--batch", batch], capture_output=True, text=True, env=dict(os.environ, **env), timeout=1800)
        m = re.search(r"Batch of \d+ queries, Is correct\?:((?: [01])+)", r.stdout)
        w = re.search(r"Batch of \d+ queries, wall \(GPU·us\): (\d+)", r.stdout)
        one = re.search(r"First dimension multiply[^:]*: (\d+)", r.stdout)
        mfma = params["nu_1"] >= 6 and params["nu_1"] <= 11 and params["nu_2"] >= 6
        if not m or not w:
            print(f"{work + ':' + variant:28s} FAILED rc={r.returncode} {r.stderr.strip()[-200:]}")
            continue
        us = int(w.group(1))
        print(f"{work + ':' + variant:28s} {params['nu_1']:3d} {params['nu_2']:3d} {'yes' if mfma else 'no':>12s} {m.group(1).strip():>16s} {us:9d} {int(batch) * 1e6 / us:9.1f} {one.group(1) if one else '-':>16s}", flush=True)

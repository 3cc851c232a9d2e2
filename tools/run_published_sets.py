#!/usr/bin/env python3
"""run every published parameter set of spiral_amd.scheme.PUBLISHED through ./spiral and print one line each
usage: python tools/run_published_sets.py [--seed N]"""
import json, os, subprocess, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from spiral_amd import scheme

seed = sys.argv[sys.argv.index("--seed") + 1] if "--seed" in sys.argv else "3"
print(f"{'set':34s} {'nu1':>3s} {'nu2':>3s} {'n':>2s} ok {'answer_us':>10s} {'sweep_us':>9s} {'db GB':>7s} {'sweep TB/s':>10s}")
for work, variants in scheme.PUBLISHED.items():
    for variant, params in variants.items():
        r = subprocess.run([sys.executable, "-m", "spiral_amd.scheme", "--set", f"{work}:{variant}", "--seed", seed], capture_output=True, text=True, timeout=1800)
        try:
            d = json.loads(r.stdout.strip().splitlines()[-1])
        except Exception:
            print(f"{work + ':' + variant:34s} FAILED {r.stderr.strip()[-200:]}")
            continue
        n = params.get("n", 2)
        db_gb = 2 ** (params["nu_1"] + params["nu_2"]) * n * n * 2048 * 8 / 1e9  # NTT-form database, reference layout
        print(f"{work + ':' + variant:34s} {params['nu_1']:3d} {params['nu_2']:3d} {n:2d} {int(d['is_corr']):2d} {d['gpu_answer_us']:10.0f} {d['gpu_sweep_us']:9.0f} {db_gb:7.2f} {db_gb / d['gpu_sweep_us'] * 1e3:10.2f}")

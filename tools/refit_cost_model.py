"""re-measure the fitting grids of spiral_amd/scheme.py on this GPU and write the coefficient file to gpurun_out/cost_model_mi355x.json (copy it over
spiral_amd/cost_model_mi355x.json afterwards); then the deviation of --select's prediction from a measured run for two workloads.  usage (through gpurun): python tools/refit_cost_model.py"""
import json, os, shutil, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (before the library initialises the device)
from spiral_amd import scheme as S

out = os.path.join(ROOT, "gpurun_out", "cost_model_mi355x.json")
os.makedirs(os.path.dirname(out), exist_ok=True)
shutil.copy(S.MODEL_PATH, out)
m = S.fit_model(out)
print("base fit error:", json.dumps(m["fit_error"]))
m = S.fit_pack_model(out)
print("pack fit error:", json.dumps(m["fit_error"]))
shutil.copy(out, S.MODEL_PATH)  # (on the GPU box's copy of the tree: the --select runs below use the new coefficients)
for argv in (["--select", "20,256", "--run", "--analyze-deviation"], ["--select", "18,30000", "--variant", "spiral-pack", "--run", "--analyze-deviation"]):
    r = subprocess.run([sys.executable, "-m", "spiral_amd.scheme"] + argv, capture_output=True, text=True, cwd=ROOT, timeout=1800)
    print(" ".join(argv), "->", r.stdout.strip().splitlines()[-1][:1500] if r.stdout.strip() else r.stderr[-500:])

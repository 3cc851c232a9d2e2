"""Run one Spiral parameter set through the ./spiral command line and print the result as one JSON object.

This is the measuring half of the reference's experiment driver (select_params.py:376-576): same keys, same
derived quantities (item_sz, dbsize, tput, rate, cost), same "factor" convention (an item larger than one
plaintext is served by `factor` database instances; the database-dependent times and the response size scale
by it).  The searching half (the cost-model fit over all_params*.pkl) is out of scope (DESIGN.md section 1, row f):
a parameter set is given explicitly (--params JSON) or picked from PUBLISHED, the sets the reference's
all_parameter_choices.txt records for its paper figures.  The scheme parameters are run-time arguments of this
build's ./spiral (environment TEXP, TEXPRIGHT, TCONV, TGSW, QPBITS, PVALUE, OUTN), where the reference
recompiles per set (select_params.py:355-371).

    python -m spiral_amd.scheme --set "20,256:spiral" --trials 3
    python -m spiral_amd.scheme --params '{"nu_1":8,"nu_2":7,"p":256,"q_prime_bits":20,"t_GSW":8,"t_conv":4,"t_exp":8,"t_exp_right":56}' --item-size 8192
"""
import argparse
import json
import math
import os
import random
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(ROOT, "spiral")

USD_PER_US = 5.41666667e-12  # select_params.py cost model: CPU time
USD_PER_BYTE = 9e-11         #                              download
POLY_LEN = 2048

# (log2 of the item count, item bytes) or a named workload -> variant -> parameters (all_parameter_choices.txt)
_BASE = {"t_exp_right": 56}
PUBLISHED = {
    "20,256": {
        "spiral": dict(_BASE, nu_1=8, nu_2=7, p=256, q_prime_bits=20, t_GSW=8, t_conv=4, t_exp=8),
        "spiral-pack": dict(_BASE, n=2, nu_1=9, nu_2=6, p=256, q_prime_bits=20, t_GSW=8, t_conv=4, t_exp=8),
        "spiralstream": dict(_BASE, direct=1, nu_1=9, nu_2=6, p=256, q_prime_bits=19, t_GSW=5, t_conv=4, t_exp=2),
        "spiralstream-pack": dict(_BASE, direct=1, n=4, nu_1=10, nu_2=3, p=1024, q_prime_bits=21, t_GSW=2, t_conv=56, t_exp=56),
    },
    "18,30000": {
        "spiral": dict(_BASE, nu_1=9, nu_2=9, p=256, q_prime_bits=21, t_GSW=9, t_conv=4, t_exp=8),
        "spiral-pack": dict(_BASE, n=4, nu_1=10, nu_2=8, p=256, q_prime_bits=20, t_GSW=8, t_conv=4, t_exp=16),
        "spiralstream": dict(_BASE, direct=1, nu_1=10, nu_2=8, p=32768, q_prime_bits=27, t_GSW=4, t_conv=32, t_exp=2),
        "spiralstream-pack": dict(_BASE, direct=1, n=4, nu_1=11, nu_2=6, p=32768, q_prime_bits=26, t_GSW=3, t_conv=56, t_exp=56),
    },
    "14,100000": {
        "spiral": dict(_BASE, nu_1=9, nu_2=5, p=512, q_prime_bits=21, t_GSW=9, t_conv=4, t_exp=16),
        "spiral-pack": dict(_BASE, n=8, nu_1=10, nu_2=4, p=128, q_prime_bits=19, t_GSW=6, t_conv=32, t_exp=8),
        "spiralstream": dict(_BASE, direct=1, nu_1=9, nu_2=5, p=16384, q_prime_bits=26, t_GSW=4, t_conv=16, t_exp=2),
        "spiralstream-pack": dict(_BASE, direct=1, n=5, nu_1=11, nu_2=3, p=65536, q_prime_bits=27, t_GSW=3, t_conv=56, t_exp=56),
    },
    "20,100000": {  # BASELINE.json configs[3]: 2^20 x 100 KB; the two variants whose database (64 GiB, 7 instances by `factor`) fits one GPU
        "spiral": dict(_BASE, nu_1=9, nu_2=11, p=256, q_prime_bits=20, t_GSW=10, t_conv=56, t_exp=16),
        "spiralstream": dict(_BASE, direct=1, nu_1=11, nu_2=9, p=32768, q_prime_bits=27, t_GSW=4, t_conv=56, t_exp=2),
    },
    "wiki": {
        "spiral": dict(_BASE, nu_1=9, nu_2=11, p=256, q_prime_bits=22, t_GSW=10, t_conv=4, t_exp=8),
        "spiral-pack": dict(_BASE, n=8, nu_1=10, nu_2=8, p=256, q_prime_bits=20, t_GSW=8, t_conv=4, t_exp=16),
        "spiralstream": dict(_BASE, direct=1, nu_1=10, nu_2=10, p=32768, q_prime_bits=27, t_GSW=4, t_conv=32, t_exp=2),
        "spiralstream-pack": dict(_BASE, direct=1, n=5, nu_1=12, nu_2=6, p=524288, q_prime_bits=31, t_GSW=3, t_conv=56, t_exp=56),
    },
    "movie": {
        "spiralstream": dict(_BASE, direct=1, nu_1=11, nu_2=3, p=32768, q_prime_bits=27, t_GSW=4, t_conv=56, t_exp=2),
        "spiralstream-pack": dict(_BASE, direct=1, n=12, nu_1=11, nu_2=3, p=524288, q_prime_bits=31, t_GSW=3, t_conv=56, t_exp=56),
    },
}
ITEM_BYTES = {"20,256": 256, "18,30000": 30000, "14,100000": 100000, "20,100000": 100000, "wiki": 30000, "movie": 2_000_000_000}

# select_params.py:386-401: what the reference scrapes from the executable's summary
SCRAPE = {
    "exp_us": r"\s+Main expansion.*:\s+([0-9]+)",
    "exp_specific_us": r"\s+Expansion took.*:\s+([0-9e\+\.]+)",
    "conv_us": r"\s+Conversion.*:\s+([0-9]+)",
    "scaltomat_us": r"\s+ScalToMat took.*:\s+([0-9]+)",
    "regtogsw_us": r"\s+RegevToGSW took.*:\s+([0-9]+)",
    "fdim_us": r"\s+First dimension multiply.*:\s+([0-9]+)",
    "fold_us": r"\s+Folding.*:\s+([0-9]+)",
    "pack_us": r"\s+Packing.*:\s+([0-9]+)",
    "query_gen_us": r"\s+Query generation.*:\s+([0-9]+)",
    "key_gen_us": r"\s+Key generation.*:\s+([0-9]+)",
    "decoding_us": r"\s+Decoding.*:\s+([0-9]+)",
    "resp_sz": r"\s+Response size.*:\s+([0-9]+)",
    "query_sz": r"\s+online query size.*:\s+([0-9]+)",
    "param_sz": r"\s+offline query size.*:\s+([0-9]+)",
    "is_corr": r"\s+Is correct?.*:\s+([0-9])",
}
GPU_EXTRAS = {
    "gpu_sweep_us": r"\s+Sweep kernels? alone.*:\s+([0-9\.]+)",
    "gpu_sweep_gbs": r"\s+Sweep kernels? alone.*\(([0-9\.]+) GB/s\)",
    "gpu_answer_us": r"\s+Whole answer, device.*:\s+([0-9\.]+)",
}


def is_high_rate(params):
    return "n" in params


def base_item_bytes(params):
    n = params.get("n", 2)
    return n * n * POLY_LEN * math.log2(params["p"]) / 8


def command(params, idx, corr=True, seed=None):
    """argv and environment of one ./spiral run for `params` (the reference's make_for + run_spiral)."""
    argv = [BIN, str(params["nu_1"]), str(params["nu_2"]), str(idx), "a"]
    if not corr:
        argv.append("--random-data")
    if is_high_rate(params):
        argv.append("--high-rate")
    if params.get("direct"):
        argv.append("--direct-upload")
    if seed is not None:
        argv += ["--seed", str(seed)]
    env = {"TEXP": params["t_exp"], "TEXPRIGHT": params["t_exp_right"], "TCONV": params["t_conv"], "TGSW": params["t_GSW"],
           "QPBITS": params["q_prime_bits"], "PVALUE": params["p"], "OUTN": params.get("n", 2)}
    return argv, {k: str(v) for k, v in env.items()}


def analyze(stdout, params, factor=1):
    """select_params.py:analyze_spiral on this build's output; GPU extras where the summary prints them."""
    hr = is_high_rate(params)
    out = {}
    for k, rx in SCRAPE.items():
        m = re.search(rx, stdout)
        if m is None:
            if k == "pack_us" and not hr:
                out[k] = 0
                continue
            raise ValueError(f"no '{k}' line in the ./spiral summary")
        out[k] = int(float(m.group(1)))
    if not hr:
        out["pack_us"] = 0
        out["query_sz"] = out["query_sz"] / 2  # the seed trick (select_params.py:423-424)
    for k in ("fdim_us", "fold_us", "pack_us", "resp_sz"):
        out[k] *= factor
    out["is_corr"] = out["is_corr"] == 1
    out["total_us"] = out["exp_us"] + out["conv_us"] + out["fdim_us"] + out["fold_us"] + out["pack_us"]
    for k, rx in GPU_EXTRAS.items():
        m = re.search(rx, stdout)
        if m:
            out[k] = float(m.group(1))
    return out


def run_once(params, factor=1, corr=True, seed=None, timeout=1800):
    idx = random.randrange(1 << (params["nu_1"] + params["nu_2"]))
    argv, env = command(params, idx, corr, seed)
    r = subprocess.run(argv, capture_output=True, text=True, env=dict(os.environ, **env), timeout=timeout)
    if r.returncode != 0:
        raise RuntimeError(f"{' '.join(argv)} exited {r.returncode}: {r.stderr.strip()[-500:]}")
    return analyze(r.stdout, params, factor)


def summarize(runs, params, item_size, factor):
    """Average the trials and add the derived columns (select_params.py:560-576)."""
    avg = {k: sum(r[k] for r in runs) / len(runs) for k in runs[0]}
    avg["item_sz"] = factor * base_item_bytes(params)
    avg["dbsize"] = avg["item_sz"] * 2 ** (params["nu_1"] + params["nu_2"])
    avg["params"] = params
    avg["tput"] = avg["dbsize"] / (avg["fdim_us"] + avg["fold_us"] + avg["pack_us"])  # bytes/us = MB/s, expansion excluded
    avg["rate"] = item_size / avg["resp_sz"]
    avg["cost"] = USD_PER_US * avg["total_us"] + USD_PER_BYTE * avg["resp_sz"]
    return avg


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("--set", help='published set "<workload>:<variant>", e.g. "20,256:spiral" or "wiki:spiralstream-pack"')
    g.add_argument("--params", help="JSON object with nu_1 nu_2 p q_prime_bits t_GSW t_conv t_exp t_exp_right [n] [direct]")
    g.add_argument("--list", action="store_true", help="print the published sets and exit")
    ap.add_argument("--item-size", type=int, help="bytes per item (default: the workload's, or one plaintext)")
    ap.add_argument("--trials", type=int, default=1)
    ap.add_argument("--random-data", action="store_true", help="pseudo-random database words, no correctness check (the reference's default)")
    ap.add_argument("--seed", type=int)
    ap.add_argument("--dry-run", action="store_true", help="print the command and exit")
    a = ap.parse_args(argv)
    if a.list:
        print(json.dumps(PUBLISHED, indent=1))
        return 0
    if a.set:
        work, _, variant = a.set.partition(":")
        try:
            params = dict(PUBLISHED[work][variant])
        except KeyError:
            ap.error(f"unknown set {a.set!r}; --list shows the published ones")
        item_size = a.item_size or ITEM_BYTES[work]
    else:
        params = json.loads(a.params)
        item_size = a.item_size or int(base_item_bytes(params))
    factor = max(1, math.ceil(item_size / base_item_bytes(params)))  # select_params.py:297-298
    if a.seed is not None:
        random.seed(a.seed)
    if a.dry_run:
        cmd, env = command(params, 0, not a.random_data, a.seed)
        print(" ".join(f"{k}={v}" for k, v in env.items()), " ".join(cmd))
        return 0
    if not os.path.exists(BIN):
        sys.exit("spiral_amd/spiral is not built: run `python -c 'import __graft_entry__ as g; g.build()'`")
    runs = [run_once(params, factor, not a.random_data, a.seed) for _ in range(a.trials)]
    print(json.dumps(summarize(runs, params, item_size, factor)))
    return 0


if __name__ == "__main__":
    sys.exit(main())

"""The reference's experiment driver (select_params.py) for this build: run a parameter set through ./spiral and print the
result as one JSON object, choose a parameter set for a database with a cost model fitted on the MI355X, or fit that model.

Measuring (select_params.py:376-576): same keys, same derived quantities (item_sz, dbsize, tput, rate, cost), same "factor"
convention (an item larger than one plaintext is served by `factor` database instances; the database-dependent times and the
response size scale by it).  The scheme parameters are run-time arguments of this build's ./spiral (environment TEXP,
TEXPRIGHT, TCONV, TGSW, QPBITS, PVALUE, OUTN), where the reference recompiles per set (select_params.py:355-371).

Selecting (select_params.py:153-198, 305, 524-540 and generate_all_schemes.py): the reference ranks a pickled table of
noise-feasible parameter sets (git-LFS objects that are not in the repository) with stage-time regressions taken on a
c5n.2xlarge.  Here the feasible sets are enumerated on the fly with a restatement of the reference's noise model
(`feasible`, pinned to outputs of generate_all_schemes.py in tests/golden/scheme_model.json) and ranked with the same cost
formula (USD per CPU-microsecond and per byte, select_params.py:119-120) over stage times predicted by a model fitted to
measurements on the MI355X (`--fit`, coefficients in spiral_amd/cost_model_mi355x.json).

    python -m spiral_amd.scheme --set "20,256:spiral" --trials 3
    python -m spiral_amd.scheme --params '{"nu_1":8,"nu_2":7,"p":256,"q_prime_bits":20,"t_GSW":8,"t_conv":4,"t_exp":8,"t_exp_right":56}' --item-size 8192
    python -m spiral_amd.scheme --select 20,256 [--variant spiral|spiralstream|spiral-pack|spiralstream-pack] [--run]
    python -m spiral_amd.scheme --fit            (on an MI355X; rewrites the coefficient file)
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
    "20,100000": {  # BASELINE.json configs[3]: 2^20 x 100 KB (the table's "Streaming" section, key "20"); the two variants whose database (64 GiB, 7 instances by `factor`) fits one GPU
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


# ---- the reference's noise model, restated (generate_all_schemes.py:9-190, 192-289) ------------------------------------------
REAL_Q = 66974689739603969      # generate_all_schemes.py:10
SIGMA, P_ERR_BITS, C_BOUND = 6.4, 40.0, 5
# plaintext moduli above 2^16 are not powers of two (generate_all_schemes.py:143-158): p -> the modulus really used
P_MOD_TABLE = {17: 131072, 18: 262144, 19: 524288, 20: 1048576, 21: 2097152, 22: 4194304, 23: 8388592, 24: 16777184, 25: 33554332,
               26: 67108804, 27: 134217608, 28: 268435216, 29: 536742296, 30: 1073612276}


def real_p(p):
    b = int(round(math.log2(p)))
    if (1 << b) != p or b < 1 or b > 30:
        raise KeyError(p)
    return p if b <= 16 else P_MOD_TABLE[b]


def noise_variance(kind, p, t_GSW, t_conv, t_exp, t_exp_right, nu_1, nu_2, n=2):
    """variance of the response's error term: calc_fast (generate_all_schemes.py:15-81) for Spiral / SpiralStream,
    calc_fast_highrate (:96-140) for the packing variants; `kind` in spiral, spiralstream, spiral-pack, spiralstream-pack"""
    q, d, sigma = float(REAL_Q), POLY_LEN, SIGMA
    z = lambda t: math.ceil(q ** (1.0 / t))
    z_GSW, z_conv, z_exp, z_exp_right = z(t_GSW), z(t_conv), z(t_exp), z(t_exp_right)
    stream = kind.startswith("spiralstream")
    if not kind.endswith("-pack"):
        m_GSW = 3 * t_GSW
        noise_scale_GSW = 4 * (t_GSW * nu_2 + 1) ** 2
        if stream:  # du_first_dim and kinda_direct_upload
            sigma_hat_regev_2 = sigma ** 2
            sigma_hat_GSW_2 = sigma ** 2
        else:
            sigma_hat_regev_2 = 4 ** (nu_1 + 1) * sigma ** 2 * (1 + d * t_exp * z_exp ** 2 / 3)
            sigma_hat_GSW_2 = noise_scale_GSW * sigma ** 2 * (1 + t_exp_right * d * z_exp_right ** 2 / 3)
        sigma_regev_2 = sigma_hat_regev_2 + d * t_conv * (z_conv ** 2) * (sigma ** 2) / 4.0
        sigma_GSW_2 = sigma_hat_GSW_2 * d * (C_BOUND * sigma) ** 2 + t_conv * d * sigma ** 2 * z_conv ** 2 / 2
        sigma_0_2 = 2 ** nu_1 * 2 * d * (p / 2) ** 2 * sigma_regev_2
        return sigma_0_2 + nu_2 * d * m_GSW * z_GSW ** 2 / 2 * sigma_GSW_2
    m_GSW = 2 * t_GSW
    sigma_regev_2 = sigma_GSW_2 = sigma ** 2
    if not stream:
        noise_scale_GSW = 4 ** (math.ceil(math.log(t_GSW * nu_2, 2)) + 1)
        sigma_regev_2 = 4 ** (nu_1 + 1) * sigma ** 2 * (1 + d * t_exp * z_exp ** 2 / 3)
        sigma_GSW_2 = noise_scale_GSW * sigma ** 2 * (1 + t_exp_right * d * z_exp_right ** 2 / 3)
        sigma_GSW_2 = sigma_GSW_2 * d * (C_BOUND * sigma) ** 2 + t_conv * d * sigma ** 2 * z_conv ** 2 / 2
    sigma_0_2 = 2 ** nu_1 * 1 * d * (p / 2) ** 2 * sigma_regev_2
    sigma_r_2 = sigma_0_2 + nu_2 * d * m_GSW * z_GSW ** 2 / 2 * sigma_GSW_2
    return sigma_r_2 + (d * n * t_conv) * (sigma ** 2) * (z_conv ** 2) / 4


def p_err_log2(p, q_prime, s_e, n=2):
    """log2 of the probability that some coefficient of the response decodes wrongly (get_p_err_fast_highrate, :159-190)"""
    pr, q, d = float(real_p(int(p))), float(REAL_Q), POLY_LEN
    thresh = 0.25 - (1.0 / 8.0) * ((4 * pr) * (q % pr) / q)
    assert 0 < thresh <= 0.25
    denom = float(s_e) * (pr / q) ** 2 + ((SIGMA ** 2) * d / 4) * (pr / float(q_prime)) ** 2
    return (math.log(2) + (-math.pi * thresh ** 2) / denom + math.log(n * n * d)) * math.log(math.e, 2)


def feasible(kind, p, t_GSW, t_conv, t_exp, t_exp_right, nu_1, nu_2, n=2):
    """None when the set cannot reach 2^-40 correctness, else {"q_prime_bits", "s_e"}: simul / simul_highrate
    (generate_all_schemes.py:192-289) -- feasibility at q' = p 2^20, then the smallest q' on the reference's ladder"""
    pack = kind.endswith("-pack")
    kw = dict(kind=kind, p=p, t_GSW=t_GSW, t_conv=t_conv, t_exp=t_exp, t_exp_right=t_exp_right, nu_1=nu_1, nu_2=nu_2, n=n)
    s_e = noise_variance(**kw)
    nn = n if pack else 2
    if p_err_log2(p, p * 2 ** 20, s_e, nn) > -P_ERR_BITS:
        return None
    bits, q_prime = (6, None) if pack else (8, None)
    while bits <= 20:
        q_prime = p * (2 ** bits) if pack else p * (2 ** bits) - p + 1
        if p_err_log2(p, q_prime, s_e, nn) <= -P_ERR_BITS:
            break
        bits += 0.1
    return {"q_prime_bits": int(math.ceil(math.log(float(q_prime), 2))), "s_e": math.log(s_e, 2)}


# ---- stage-time model fitted on the MI355X (the analogue of select_params.py:179-187) ------------------------------------------
MODEL_PATH = os.path.join(ROOT, "cost_model_mi355x.json")
MIN_Q_PRIME_BITS = 14  # select_params.py:121
FIT_GRID = [  # (nu_1, nu_2, t_GSW, t_conv, t_exp, direct): spans the published sets' ranges
    (6, 4, 8, 4, 8, 0), (7, 5, 8, 4, 8, 0), (8, 6, 8, 4, 8, 0), (8, 7, 8, 4, 8, 0), (9, 6, 8, 4, 8, 0), (9, 7, 8, 4, 8, 0), (9, 8, 8, 4, 8, 0),
    (9, 9, 9, 4, 8, 0), (10, 7, 8, 4, 8, 0), (10, 9, 8, 4, 8, 0), (8, 8, 4, 4, 8, 0), (8, 8, 6, 4, 8, 0), (8, 8, 10, 4, 8, 0), (8, 8, 12, 4, 8, 0),
    (9, 5, 9, 4, 16, 0), (8, 7, 8, 4, 4, 0), (8, 7, 8, 4, 16, 0), (8, 7, 8, 4, 32, 0), (8, 7, 8, 8, 8, 0), (8, 7, 8, 16, 8, 0), (9, 10, 10, 4, 8, 0),
    (7, 9, 8, 4, 8, 0), (10, 5, 6, 4, 8, 0), (6, 9, 8, 8, 16, 0), (9, 11, 10, 4, 8, 0),
    (9, 6, 5, 4, 2, 1), (10, 8, 4, 32, 2, 1), (9, 5, 4, 16, 2, 1), (11, 9, 4, 56, 2, 1), (10, 10, 4, 32, 2, 1), (11, 3, 4, 56, 2, 1), (8, 6, 5, 4, 2, 1),
    (10, 6, 6, 8, 2, 1), (11, 7, 4, 16, 2, 1), (9, 9, 8, 4, 2, 1),
]


def _ceil_log2(x):
    r = 0
    while (1 << r) < x:
        r += 1
    return r


def model_features(nu_1, nu_2, t_GSW, t_conv, t_exp, t_exp_right=56, direct=0):
    """what the stage times are regressed on: counts of limb-pair transforms, of products, of launches and of database bytes,
    derived from the parameters exactly as the server schedules them (host_common.h run_expand, server.cpp)"""
    dim0, num_per, ell = 1 << nu_1, 1 << nu_2, t_GSW
    f = {"db_words": dim0 * num_per * 4 * POLY_LEN, "lift_polys": num_per * 6}
    if direct:
        f.update(exp_rounds=0, exp_transforms=0, exp_macs=0)
    else:
        g = _ceil_log2(dim0 + ell * nu_2)
        stop = _ceil_log2(ell * nu_2) if nu_2 and ell * nu_2 <= dim0 else 0
        tr = mac = 0
        for r in range(g):
            even = 1 << r
            odd = 0 if (stop and r > stop) else (min(1 << r, ell * nu_2 + 1) if (stop and r == stop) else (1 << r))
            tr += even * (1 + t_exp) + odd * (1 + t_exp_right)
            mac += even * t_exp + odd * t_exp_right
        f.update(exp_rounds=g, exp_transforms=tr, exp_macs=mac)
    f["conv_transforms"] = (dim0 + 2 * nu_2 * ell) * (1 + t_conv)
    f["conv_macs"] = dim0 * 6 * t_conv + nu_2 * ell * (6 * t_conv + 6 * t_conv)
    f["fold_rounds"] = nu_2
    # pair form of a fold round: per pair of ciphertexts 6 polynomials x (2 lifts + ell digit-difference transforms) and a
    # product of m2 = 3 ell terms per output polynomial; summed over the rounds (np' = num_per/2 ... 1 pairs)
    f["fold_transforms"] = (num_per - 1) * 6 * (ell + 2) if nu_2 else 0
    f["fold_macs"] = (num_per - 1) * 6 * 3 * ell
    return f


# stage -> the features it is regressed on (plus a constant); times in microseconds
MODEL_TERMS = {
    "exp_us": ["exp_rounds", "exp_transforms", "exp_macs"],
    "conv_us": ["conv_transforms", "conv_macs"],
    "fdim_us": ["db_words", "lift_polys"],
    "fold_us": ["fold_rounds", "fold_transforms", "fold_macs"],
}


# SpiralPack / SpiralStreamPack: the same regression on what pack_server.cpp schedules; (nu_1, nu_2, t_GSW, t_conv, t_exp, n, direct).
# The database is n^2 trial images of 2^(nu_1+nu_2) single polynomials (7 bytes per word on the device): the grid stays under 100 GiB.
PACK_FIT_GRID = [
    (9, 6, 8, 4, 8, 2, 0), (9, 6, 8, 4, 8, 4, 0), (10, 8, 8, 4, 16, 4, 0), (10, 4, 6, 32, 8, 8, 0), (10, 8, 8, 4, 16, 2, 0), (8, 6, 8, 4, 8, 2, 0), (9, 7, 8, 4, 8, 2, 0),
    (9, 5, 6, 8, 8, 4, 0), (10, 6, 8, 4, 8, 4, 0), (9, 8, 10, 4, 16, 2, 0), (8, 7, 4, 16, 4, 4, 0), (9, 4, 8, 4, 32, 8, 0), (10, 5, 5, 4, 8, 12, 0), (9, 6, 12, 4, 8, 2, 0),
    (7, 7, 8, 8, 8, 4, 0), (10, 7, 9, 4, 8, 2, 0), (9, 9, 8, 4, 8, 2, 0), (8, 8, 6, 4, 16, 4, 0),
    (10, 3, 2, 56, 56, 4, 1), (11, 6, 3, 56, 56, 4, 1), (11, 3, 3, 56, 56, 5, 1), (12, 6, 3, 56, 56, 5, 1), (11, 3, 3, 56, 56, 12, 1), (10, 5, 4, 56, 56, 6, 1),
    (11, 5, 2, 56, 56, 8, 1), (9, 6, 5, 56, 56, 4, 1), (12, 4, 3, 56, 56, 7, 1), (10, 7, 3, 56, 56, 4, 1), (11, 7, 4, 56, 56, 4, 1), (9, 9, 6, 56, 56, 4, 1),
    (13, 3, 3, 56, 56, 5, 1), (12, 2, 3, 56, 56, 5, 1), (13, 2, 4, 56, 56, 4, 1), (12, 3, 2, 56, 56, 9, 1), (13, 4, 3, 56, 56, 4, 1), (10, 2, 8, 4, 8, 4, 0), (10, 3, 6, 8, 16, 8, 0),
    (8, 2, 6, 4, 8, 12, 0), (8, 4, 16, 4, 56, 12, 0), (9, 3, 8, 4, 16, 12, 0), (9, 5, 8, 8, 8, 8, 0), (8, 5, 10, 4, 32, 8, 0), (9, 4, 12, 16, 4, 12, 0), (7, 6, 14, 4, 16, 8, 0),
]


def pack_model_features(nu_1, nu_2, t_GSW, t_conv, t_exp, n, t_exp_right=56, direct=0):
    """counts behind the stage times of the packing variants, as pack_server.cpp schedules them (n^2 trials batched)"""
    dim0, num_per, ell, trials = 1 << nu_1, 1 << nu_2, t_GSW, n * n
    f = {"db_words": trials * dim0 * num_per * POLY_LEN, "acc_polys": trials * num_per * 2, "trials": trials,
         "query_words": trials * dim0 * 2 * POLY_LEN}  # the first-dimension query records, streamed once per trial
    if direct:
        f.update(exp_rounds=0, exp_transforms=0, exp_macs=0, conv_transforms=0, conv_macs=0)
    else:
        base = model_features(nu_1, nu_2, t_GSW, t_conv, t_exp, t_exp_right, 0)
        f.update(exp_rounds=base["exp_rounds"], exp_transforms=base["exp_transforms"], exp_macs=base["exp_macs"])
        f["conv_transforms"] = 2 * nu_2 * ell * (1 + t_conv)  # regevToSimpleGsw: both rows of every GSW-bit ciphertext
        f["conv_macs"] = nu_2 * ell * 2 * 2 * t_conv
    f["key_polys"] = nu_2 * 2 * 4 * ell  # fold keys assembled from the (uploaded or converted) GSW ciphertexts
    f["fold_rounds"] = nu_2
    # pair form: per round 2 np cts x 2 polys lifted, np pairs x 2 polys x ell digit-difference transforms, a product of 2 ell terms; + the last lift
    f["fold_transforms"] = trials * ((num_per - 1) * (4 + 2 * ell) + 2)
    f["fold_macs"] = trials * (num_per - 1) * 2 * 2 * ell
    f["pack_transforms"] = trials * (1 + t_conv) + n * (n + 1)  # digits of every trial's row 0, the lift of the packed ct
    f["pack_macs"] = n * (n + 1) * n * t_conv
    return f


PACK_MODEL_TERMS = {
    "exp_us": ["exp_rounds", "exp_transforms", "exp_macs"],
    "conv_us": ["conv_transforms", "conv_macs", "key_polys"],
    "fdim_us": ["db_words", "query_words", "acc_polys", "trials"],
    "fold_us": ["fold_rounds", "fold_transforms", "fold_macs"],
    "pack_us": ["pack_transforms", "pack_macs"],
}


def is_pack(params):
    return "n" in params


def load_model(path=None):
    return json.load(open(path or MODEL_PATH))


def predict_times(model, params):
    if is_pack(params):
        f = pack_model_features(params["nu_1"], params["nu_2"], params["t_GSW"], params["t_conv"], params["t_exp"], params["n"], params.get("t_exp_right", 56), params.get("direct", 0))
        out = {}
        for stage, terms in PACK_MODEL_TERMS.items():
            c = model["pack"]["coefficients"][stage]
            out[stage] = max(0.0, c["const"] * (0 if (stage == "exp_us" and params.get("direct")) else 1) + sum(c[t] * f[t] for t in terms))
        return out
    f = model_features(params["nu_1"], params["nu_2"], params["t_GSW"], params["t_conv"], params["t_exp"], params.get("t_exp_right", 56), params.get("direct", 0))
    out = {}
    for stage, terms in MODEL_TERMS.items():
        c = model["coefficients"][stage]
        out[stage] = max(0.0, c["const"] * (0 if (stage == "exp_us" and params.get("direct")) else 1) + sum(c[t] * f[t] for t in terms))
    return out


def predicted_cost(model, params, item_size):
    """select_params.py:153-215 with the MI355X stage times: (cost in USD, total_us, response bytes, factor)"""
    n = params.get("n", 2)
    factor = max(1, math.ceil(item_size / (n * n * POLY_LEN * math.log2(real_p(params["p"])) / 8)))
    t = predict_times(model, params)
    qpb = max(params["q_prime_bits"], MIN_Q_PRIME_BITS)
    if is_pack(params):  # calc_cost_highrate (select_params.py:217-266): n^2 trials per instance (inside the fitted stage times), one packed response
        total_us = t["exp_us"] + t["conv_us"] + factor * (t["fdim_us"] + t["fold_us"] + t["pack_us"])
        total_bytes = factor * ((n * n * 2048 * math.ceil(math.log2(4 * params["p"])) + n * 2048 * qpb) / 8)
    else:
        total_us = t["exp_us"] + t["conv_us"] + factor * (t["fdim_us"] + t["fold_us"])
        total_bytes = factor * ((2 * 2 * 2048 * math.ceil(math.log2(4 * params["p"])) + 2 * 2048 * qpb) / 8)
    return USD_PER_US * total_us + USD_PER_BYTE * total_bytes, total_us, total_bytes, factor


DEVICE_DB_BUDGET = 256 << 30  # bytes of one MI355X's 288 GB left to the database image(s) of a candidate


def device_db_bytes(params, item_size):
    """what the candidate's database occupies on the device: `factor` instances (select_params.py:297) of 4 polynomials per item
    (n^2 trial images of one polynomial for the packing variants), 7 bytes per word"""
    n = params.get("n", 2)
    factor = max(1, math.ceil(item_size / (n * n * POLY_LEN * math.log2(real_p(params["p"])) / 8)))
    return factor * (n * n if is_pack(params) else 4) * (1 << (params["nu_1"] + params["nu_2"])) * POLY_LEN * 7


def enumerate_sets(variant, log_n, item_size):
    """noise-feasible parameter sets whose database holds 2^log_n items of item_size bytes: the reference's search space
    (generate_all_schemes.py get_regular_choices / get_streaming_choices) filtered by select_params.py pred (:305-330), with
    nu_1 + nu_2 kept within 1 of the smallest that fits (the cost grows with both)"""
    if variant.endswith("-pack"):
        return enumerate_pack_sets(variant, log_n, item_size)
    stream = variant == "spiralstream"
    target = (1 << log_n) * item_size
    out = []
    for pb in range(2, (20 if stream else 15) + 1):
        p = 1 << pb
        base_item = 4 * POLY_LEN * math.log2(real_p(p)) / 8
        factor = math.ceil(item_size / base_item)
        need = _ceil_log2(math.ceil(target / (factor * base_item)))
        if stream:
            need = max(need, log_n) if item_size == 1 else need
        for total_nu in (max(need, 10 if not stream else 4), max(need, 10 if not stream else 4) + 1):
            for nu_1 in range(2, (13 if stream else 10) + 1):
                nu_2 = total_nu - nu_1
                if nu_2 < 2 or nu_2 > 13:
                    continue
                for t_GSW in range(2, 29):
                    for t_conv in (2, 4, 8, 16, 32, 56):
                        for t_exp in ((2,) if stream else (2, 4, 8, 16, 32, 56)):
                            r = feasible(variant, p, t_GSW, t_conv, t_exp, 56, nu_1, nu_2)
                            if r is None:
                                continue
                            if not stream and _ceil_log2((1 << nu_1) + t_GSW * nu_2) > 11:
                                continue  # the query must fit one polynomial
                            prm = dict(nu_1=nu_1, nu_2=nu_2, p=p, q_prime_bits=max(r["q_prime_bits"], MIN_Q_PRIME_BITS), t_GSW=t_GSW, t_conv=t_conv, t_exp=t_exp,
                                       t_exp_right=56, s_e=r["s_e"])
                            if stream:
                                prm["direct"] = 1
                            out.append(prm)
    return out


def enumerate_pack_sets(variant, log_n, item_size):
    """the same for the packing variants: get_highrate_choices / get_highrate_streaming_choices (generate_all_schemes.py:
    371-421) -- n in 2, 4, 8, 12 with every t_exp, t_conv (SpiralPack); n in 4..12, p from 2^10, t_GSW <= 10, t_exp = t_conv = 56
    (SpiralStreamPack) -- filtered by select_params.py pred (:305-336)"""
    stream = variant == "spiralstream-pack"
    target = (1 << log_n) * item_size
    out = []
    for n in (range(4, 13) if stream else (2, 4, 8, 12)):
        for pb in (range(10, 31) if stream else range(2, 21)):
            p = 1 << pb
            base_item = n * n * POLY_LEN * math.log2(real_p(p)) / 8
            factor = math.ceil(item_size / base_item)
            need = max(_ceil_log2(math.ceil(target / (factor * base_item))), 10 if not stream else 4)
            for total_nu in (need, need + 1):
                for nu_1 in range(2, (13 if stream else 10) + 1):
                    nu_2 = total_nu - nu_1
                    if nu_2 < 2 or nu_2 > 13:
                        continue
                    for t_GSW in range(2, (10 if stream else 16) + 1):
                        if not stream and _ceil_log2((1 << nu_1) + t_GSW * nu_2) > 11:
                            continue  # the query must fit one polynomial
                        for t_conv in ((56,) if stream else (2, 4, 8, 16, 32, 56)):
                            for t_exp in ((56,) if stream else (2, 4, 8, 16, 32, 56)):
                                r = feasible(variant, p, t_GSW, t_conv, t_exp, 56, nu_1, nu_2, n)
                                if r is None:
                                    continue
                                prm = dict(n=n, nu_1=nu_1, nu_2=nu_2, p=p, q_prime_bits=max(r["q_prime_bits"], MIN_Q_PRIME_BITS), t_GSW=t_GSW, t_conv=t_conv,
                                           t_exp=t_exp, t_exp_right=56, s_e=r["s_e"])
                                if stream:
                                    prm["direct"] = 1
                                out.append(prm)
    return out


def select(variant, log_n, item_size, model=None, top=5, optimize_for="cost", usd_per_us=None, max_db_bytes=None):
    """the best feasible sets under the MI355X model, best first: [(cost, total_us, bytes, factor, params)].  optimize_for:
    "cost" (USD per query, select_params.py:198-205; usd_per_us overrides the reference's price of a CPU-microsecond, which
    undervalues a GPU-microsecond by two orders of magnitude), "tput" (server time) or "rate" (response size), the reference's
    --optimize-for shortcuts (select_params.py:268-276); max_db_bytes drops candidates whose device database is larger"""
    model = model or load_model()
    ranked = []
    for prm in enumerate_sets(variant, log_n, item_size):
        try:
            if max_db_bytes is not None and device_db_bytes(prm, item_size) > max_db_bytes:
                continue  # --one-gpu: the candidate's database image(s) must be resident on one device
            cost, total_us, nbytes, factor = predicted_cost(model, prm, item_size)
        except KeyError:
            continue
        if usd_per_us is not None:
            cost = usd_per_us * total_us + USD_PER_BYTE * nbytes
        key = {"cost": cost, "tput": total_us, "rate": nbytes}[optimize_for]
        ranked.append((key, (cost, total_us, nbytes, factor, prm)))
    ranked.sort(key=lambda x: x[0])
    return [r[1] for r in ranked[:top]]


def fit_model(out_path=None, reps=12):
    """measure the stage times of FIT_GRID on this GPU (resident server, arbitrary valid database words, synthetic keys and
    query: the times do not depend on the values) and regress them on model_features: writes the coefficient file"""
    import numpy as np

    import spiral_amd as sa

    rows = []
    rng = np.random.default_rng(1)
    mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
    for nu_1, nu_2, t_GSW, t_conv, t_exp, direct in FIT_GRID:
        pg = sa.make_params(nu_1, nu_2, t_gsw=t_GSW, t_conv=t_conv, t_exp=t_exp, t_exp_right=56, qprime_bits=20, p_db=256, direct_upload=direct)
        shp = sa.get_shape(pg)
        srv = sa.Server(pg)
        srv.fill_db_random(3)
        srv.set_pub_params(mk((max(shp.n_left, 1), 2, t_exp)), mk((max(shp.n_right, 1), 2, 56)), mk((3, 2 * t_conv)), mk((3, 2 * t_conv)))
        srv.set_query(mk((shp.n_query_cts, 2)))
        us = [srv.answer_resident() for _ in range(reps)][2:]
        med = {k: float(np.median([u[k] for u in us])) for k in us[0]}
        srv.close()
        rows.append({"params": dict(nu_1=nu_1, nu_2=nu_2, t_GSW=t_GSW, t_conv=t_conv, t_exp=t_exp, direct=direct),
                     "measured": {"exp_us": med["expansion_us"], "conv_us": med["conversion_us"], "fdim_us": med["first_dim_us"], "fold_us": med["folding_us"] + med["response_us"]},
                     "features": model_features(nu_1, nu_2, t_GSW, t_conv, t_exp, 56, direct)})
    coef, err = {}, {}
    for stage, terms in MODEL_TERMS.items():
        use = [r for r in rows if not (stage == "exp_us" and r["params"]["direct"])]
        A = np.array([[1.0] + [r["features"][t] for t in terms] for r in use])
        y = np.array([r["measured"][stage] for r in use])
        w = 1.0 / np.maximum(y, 20.0)  # relative error matters: a 300 us stage and a 10 ms stage both count
        x, *_ = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)
        coef[stage] = dict(zip(["const"] + terms, [float(v) for v in x]))
        pred = A @ x
        err[stage] = {"median_rel_err": float(np.median(np.abs(pred - y) / np.maximum(y, 1))), "max_rel_err": float(np.max(np.abs(pred - y) / np.maximum(y, 1)))}
    model = {"device": "MI355X (gfx950), one GPU, eager stage launches timed with HIP events (Server.answer_resident)",
             "units": "microseconds; features: spiral_amd.scheme.model_features", "coefficients": coef, "fit_error": err, "grid": rows}
    if os.path.exists(out_path or MODEL_PATH):  # the packing variants' section is fitted separately (fit_pack_model)
        old = json.load(open(out_path or MODEL_PATH))
        if "pack" in old:
            model["pack"] = old["pack"]
    json.dump(model, open(out_path or MODEL_PATH, "w"), indent=1)
    return model


def _regress(rows, terms_by_stage, np, skip=lambda stage, r: False):
    coef, err = {}, {}
    for stage, terms in terms_by_stage.items():
        use = [r for r in rows if not skip(stage, r)]
        A = np.array([[1.0] + [r["features"][t] for t in terms] for r in use])
        y = np.array([r["measured"][stage] for r in use])
        w = 1.0 / np.maximum(y, 20.0)
        x, *_ = np.linalg.lstsq(A * w[:, None], y * w, rcond=None)
        coef[stage] = dict(zip(["const"] + terms, [float(v) for v in x]))
        pred = A @ x
        err[stage] = {"median_rel_err": float(np.median(np.abs(pred - y) / np.maximum(y, 1))), "max_rel_err": float(np.max(np.abs(pred - y) / np.maximum(y, 1)))}
    return coef, err


def fit_pack_model(out_path=None, reps=6):
    """the packing variants' section of the coefficient file: PACK_FIT_GRID through PackServer.answer (its stage buckets are HIP
    events between the stages of one answer), regressed on pack_model_features"""
    import numpy as np

    import spiral_amd as sa

    rows = []
    rng = np.random.default_rng(2)
    mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
    for nu_1, nu_2, t_GSW, t_conv, t_exp, n, direct in PACK_FIT_GRID:
        pg = sa.make_params(nu_1, nu_2, t_gsw=t_GSW, t_conv=t_conv, t_exp=t_exp, t_exp_right=56, qprime_bits=20, p_db=256, direct_upload=direct)
        shp = sa.get_pack_shape(pg, n)
        srv = sa.PackServer(pg, n)
        srv.fill_db_random(3)
        srv.set_pub_params(mk((max(shp.n_left, 1), 2, t_exp)), mk((max(shp.n_right, 1), 2, 56)), mk((2, 2 * t_conv)), mk((n, n + 1, t_conv)))
        q = mk((shp.n_query_cts, 2))
        us = [srv.answer(q, want_packed=False)[2] for _ in range(reps)][2:]
        med = {k: float(np.median([u[k] for u in us])) for k in us[0]}
        srv.close()
        rows.append({"params": dict(nu_1=nu_1, nu_2=nu_2, t_GSW=t_GSW, t_conv=t_conv, t_exp=t_exp, n=n, direct=direct),
                     "measured": {"exp_us": med["expansion_us"], "conv_us": med["conversion_us"], "fdim_us": med["first_dim_us"], "fold_us": med["folding_us"],
                                  "pack_us": med["packing_us"], "total_us": med["total_us"]},
                     "features": pack_model_features(nu_1, nu_2, t_GSW, t_conv, t_exp, n, 56, direct)})
    coef, err = _regress(rows, PACK_MODEL_TERMS, np, skip=lambda stage, r: stage in ("exp_us",) and r["params"]["direct"])
    path = out_path or MODEL_PATH
    model = json.load(open(path)) if os.path.exists(path) else {}
    model["pack"] = {"units": "microseconds per answer (all n^2 trials); features: spiral_amd.scheme.pack_model_features", "coefficients": coef, "fit_error": err, "grid": rows}
    json.dump(model, open(path, "w"), indent=1)
    return model["pack"]


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
    g.add_argument("--select", metavar="LOGN,ITEMSIZE", help="choose the cheapest noise-feasible parameter set for 2^LOGN items of ITEMSIZE bytes under the MI355X cost model")
    g.add_argument("--fit", action="store_true", help="measure the fitting grid on this GPU and rewrite the cost-model coefficients")
    g.add_argument("--fit-pack", action="store_true", help="the same for the packing variants' section of the coefficient file")
    ap.add_argument("--variant", default="spiral", choices=["spiral", "spiralstream", "spiral-pack", "spiralstream-pack"],
                    help="--select: query compression (spiral) or direct upload (spiralstream), each with or without response packing")
    ap.add_argument("--run", action="store_true", help="--select: also run the chosen set through ./spiral and report the measured times")
    ap.add_argument("--analyze-deviation", action="store_true", help="--select --run: predicted against measured stage times (select_params.py --analyze-deviation)")
    ap.add_argument("--top", type=int, default=5, help="--select: how many of the cheapest sets to list")
    ap.add_argument("--optimize-for", default="cost", choices=["cost", "tput", "rate"], help="--select: USD per query (default), server time, or response size")
    ap.add_argument("--one-gpu", action="store_true", help="--select: only sets whose database image(s) fit one MI355X (256 GiB of its 288 GB)")
    ap.add_argument("--usd-per-us", type=float, help="--select: price of a server microsecond (default: the reference's CPU figure, 5.4e-12)")
    ap.add_argument("--item-size", type=int, help="bytes per item (default: the workload's, or one plaintext)")
    ap.add_argument("--trials", type=int, default=1)
    ap.add_argument("--random-data", action="store_true", help="pseudo-random database words, no correctness check (the reference's default)")
    ap.add_argument("--seed", type=int)
    ap.add_argument("--dry-run", action="store_true", help="print the command and exit")
    a = ap.parse_args(argv)
    if a.list:
        print(json.dumps(PUBLISHED, indent=1))
        return 0
    if a.fit:
        m = fit_model()
        print(json.dumps({"coefficients": m["coefficients"], "fit_error": m["fit_error"], "points": len(m["grid"]), "written": MODEL_PATH}))
        return 0
    if a.fit_pack:
        m = fit_pack_model()
        print(json.dumps({"coefficients": m["coefficients"], "fit_error": m["fit_error"], "points": len(m["grid"]), "written": MODEL_PATH}))
        return 0
    if a.select:
        log_n, item_size = (int(x) for x in a.select.split(","))
        model = load_model()
        ranked = select(a.variant, log_n, item_size, model, a.top, a.optimize_for, a.usd_per_us, DEVICE_DB_BUDGET if a.one_gpu else None)
        if not ranked:
            sys.exit("no feasible parameter set")
        out = {"workload": {"log_n": log_n, "item_size": item_size, "variant": a.variant, "optimize_for": a.optimize_for}, "model": os.path.basename(MODEL_PATH),
               "candidates": [{"params": prm, "predicted_us": predict_times(model, prm), "predicted_total_us": tot, "resp_bytes": nb, "factor": fac, "cost_usd": cost,
                               "device_db_bytes": device_db_bytes(prm, item_size), "gpus_for_db": math.ceil(device_db_bytes(prm, item_size) / DEVICE_DB_BUDGET)}
                              for cost, tot, nb, fac, prm in ranked]}
        work = f"{log_n},{item_size}"
        if work in PUBLISHED and a.variant in PUBLISHED[work]:  # what the reference's CPU model chose for the same workload, under this model
            pub = dict(PUBLISHED[work][a.variant])
            c = predicted_cost(model, pub, item_size)
            out["published_choice"] = {"params": pub, "predicted_total_us": c[1], "cost_usd": c[0]}
        if a.run:
            if not os.path.exists(BIN):
                sys.exit("spiral_amd/spiral is not built")
            best = {k: v for k, v in ranked[0][4].items() if k != "s_e"}
            runs = [run_once(best, ranked[0][3], not a.random_data, a.seed) for _ in range(max(1, a.trials))]
            out["measured"] = summarize(runs, best, item_size, ranked[0][3])
            if a.analyze_deviation:  # select_params.py:589-616: the model's stage times against the measured ones
                fac, pred, meas = ranked[0][3], predict_times(model, ranked[0][4]), out["measured"]
                keys = ["exp_us", "conv_us"] + (["pack_us"] if is_pack(best) else []) + ["fdim_us", "fold_us"]
                predicted = [pred[k] * (1 if k in ("exp_us", "conv_us") else fac) for k in keys]
                actual = [meas[k] for k in keys]
                predicted.append(sum(predicted))
                actual.append(meas["total_us"])
                out["deviation"] = {"factor": fac, "name": keys + ["total_us"], "predicted_times": predicted, "actual_times": actual,
                                    "abs_err": [abs(x - y) for x, y in zip(predicted, actual)],
                                    "rel_err": [abs(x - y) / x if x else 0 for x, y in zip(predicted, actual)]}
        print(json.dumps(out))
        if "deviation" in out:  # the reference's table, after the JSON line
            d = out["deviation"]
            print("factor", d["factor"])
            for row in ("name", "predicted_times", "actual_times", "abs_err", "rel_err"):
                print(" ".join([row] + [str(round(v, 3)) if isinstance(v, float) else str(v) for v in d[row]]) if row != "name" else " ".join(["name"] + d["name"]))
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

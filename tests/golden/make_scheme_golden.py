#!/usr/bin/env python3
"""Generates tests/golden/scheme_model.json from the reference (run in the build container, where /root/reference exists):

* `published`: the parameter sets of /root/reference/all_parameter_choices.txt (the data file the reference's paper figures
  use), verbatim -- pins spiral_amd.scheme.PUBLISHED;
* `noise`: outputs of the reference's own noise model, obtained by IMPORTING /root/reference/generate_all_schemes.py and
  calling simul_normal / simul_stream / simul_highrate_normal / simul_highrate_stream (generate_all_schemes.py:192-289) on a
  seeded sample of parameter choices: for each, null (infeasible) or the q' bit width and log2 noise variance it returns --
  pins the restated noise model of spiral_amd.scheme (feasible()).

Only data goes into the fixture; no reference source text."""
import importlib.util
import json
import os
import random
import re

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))


def published():
    txt = open(os.path.join(REF, "all_parameter_choices.txt")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    dec = json.JSONDecoder()
    out, pos = {}, 0
    while True:  # the file is a sequence of JSON objects separated by comments
        m = re.compile(r"\S").search(txt, pos)
        if not m:
            break
        obj, pos = dec.raw_decode(txt, m.start())
        for k, v in obj.items():
            out.setdefault(k, {}).update({variant: e["params"] for variant, e in v.items() if isinstance(e, dict) and "params" in e})
    return out


def noise_vectors():
    spec = importlib.util.spec_from_file_location("ref_generate_all_schemes", os.path.join(REF, "generate_all_schemes.py"))
    g = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(g)
    rng = random.Random(20261002)
    vec = []
    nus = [(j1, j2) for j1 in range(2, 12) for j2 in range(2, 12) if j1 + j2 >= 10]
    for kind, fn, hr in (("spiral", g.simul_normal, False), ("spiralstream", g.simul_stream, False),
                         ("spiral-pack", g.simul_highrate_normal, True), ("spiralstream-pack", g.simul_highrate_stream, True)):
        for _ in range(150):
            c = {"p": 2 ** rng.randrange(2, 21 if "stream" in kind or hr else 16), "q": g.real_q, "t_GSW": rng.randrange(2, 20),
                 "t_exp": rng.choice([2, 4, 8, 16, 32, 56]), "t_exp_right": 56, "t_conv": rng.choice([2, 4, 8, 16, 32, 56]), "factor": 1,
                 "nu_1,nu_2": rng.choice(nus)}
            if kind == "spiral":  # the reference also drops (t_exp, nu_1) pairs its CPU expansion LUT has no entry for (:194): keep to those it has
                c["t_exp"], nu1 = rng.choice(sorted(k for k in g.exp_lut_keys if k[0] in (2, 4, 8, 16, 32, 56) and k[1] >= 2))
                c["nu_1,nu_2"] = (nu1, rng.randrange(max(2, 10 - nu1), 12))
            if kind == "spiralstream-pack":
                c.update(t_exp=56, t_conv=56, t_GSW=rng.randrange(2, 11), p=2 ** rng.randrange(10, 25))
            sel = [c[k] for k in g.ks]
            if hr:
                sel.append(rng.choice([2, 4, 8, 12] if kind == "spiral-pack" else [4, 5, 8, 12]))
            try:
                r = fn(sel)
            except (KeyError, AssertionError):  # p outside the reference's table of plaintext moduli
                continue
            entry = {"kind": kind, "p": c["p"], "t_GSW": c["t_GSW"], "t_exp": c["t_exp"], "t_exp_right": 56, "t_conv": c["t_conv"],
                     "nu_1": c["nu_1,nu_2"][0], "nu_2": c["nu_1,nu_2"][1]}
            if hr:
                entry["n"] = sel[-1]
            if r is None:
                entry["feasible"] = False
            else:
                cl = g.clean(r)
                entry.update(feasible=True, q_prime_bits=cl["q_prime_bits"], s_e=cl["s_e"])
            vec.append(entry)
    return vec


if __name__ == "__main__":
    out = {"source": "menonsamir/spiral all_parameter_choices.txt + outputs of generate_all_schemes.py simul_* (imported, seeded sample)",
           "published": published(), "noise": noise_vectors()}
    json.dump(out, open(os.path.join(HERE, "scheme_model.json"), "w"), indent=0, sort_keys=True)
    print(len(out["published"]), "workloads,", len(out["noise"]), "noise vectors,", sum(v["feasible"] for v in out["noise"]), "feasible")

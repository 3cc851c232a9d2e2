"""The ./spiral drop-in command line: same argv and the same scraped stdout lines as the reference executable
(src/spiral.cpp:1242-1303, 209-265; regexes of select_params.py:386-401), its own host client, HIP server path."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "spiral_amd", "spiral")

# select_params.py:386-401, verbatim
REGEXES = {
    "exp_us": r"\s+Main expansion.*:\s+([0-9]+)",
    "exp_specific_us": r"\s+Expansion took.*:\s+([0-9e\+\.]+)",
    "conv_us": r"\s+Conversion.*:\s+([0-9]+)",
    "scaltomat_us": r"\s+ScalToMat took.*:\s+([0-9]+)",
    "regtogsw_us": r"\s+RegevToGSW took.*:\s+([0-9]+)",
    "fdim_us": r"\s+First dimension multiply.*:\s+([0-9]+)",
    "fold_us": r"\s+Folding.*:\s+([0-9]+)",
    "query_gen_us": r"\s+Query generation.*:\s+([0-9]+)",
    "key_gen_us": r"\s+Key generation.*:\s+([0-9]+)",
    "decoding_us": r"\s+Decoding.*:\s+([0-9]+)",
    "resp_sz": r"\s+Response size.*:\s+([0-9]+)",
    "query_sz": r"\s+online query size.*:\s+([0-9]+)",
    "param_sz": r"\s+offline query size.*:\s+([0-9]+)",
    "is_corr": r"\s+Is correct?.*:\s+([0-9])",
}


def test_cli_refuses_to_run_without_gpu_or_args():
    assert os.path.exists(BIN), "build() must produce spiral_amd/spiral"
    r = subprocess.run([BIN], capture_output=True, text=True)
    assert r.returncode == 1 and "usage" in r.stderr
    r = subprocess.run([BIN, "4", "2", "999999", "a"], capture_output=True, text=True)
    assert r.returncode == 1 and "out of range" in r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize("args,env", [
    (["4", "2", "37", "a", "--seed", "3"], {"TGSW": "4"}),
    (["4", "2", "21", "a", "--output-err", "/tmp/spiral_err_not_written.txt", "--seed", "9"], {"TGSW": "4"}),  # argv of src/spiral.cpp:1287-1291: consumed, ignored
    (["6", "2", "77", "a", "--random-data", "--seed", "4"], {}),
    (["5", "2", "7", "a", "--direct-upload", "--seed", "5"], {"TEXP": "2", "TGSW": "5", "QPBITS": "19"}),  # SURVEY 8c stream probe
    (["5", "6", "1234", "a", "--seed", "11", "--batch", "4"], {}),  # + four clients answered by one spiral_gpu_server_run_query_batch call (C++ consumer of the batch entry point)
    (["4", "2", "3", "a", "--direct-upload", "--seed", "12", "--batch", "3"], {"TEXP": "2", "TGSW": "5", "QPBITS": "19"}),
    (["5", "6", "99", "a", "--seed", "13", "--batch", "7"], {}),
    # an item of three plaintexts = three database instances, one query (the SpiralStream form and with query compression): C++ consumer of answer_instances
    (["5", "2", "7", "a", "--direct-upload", "--seed", "5", "--instances", "3"], {"TEXP": "2", "TGSW": "5", "QPBITS": "19"}),
    (["4", "3", "40", "a", "--seed", "14", "--instances", "4"], {}),
])
def test_cli_end_to_end(args, env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([BIN] + args, capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    got = {k: re.search(rx, r.stdout) for k, rx in REGEXES.items()}
    missing = [k for k, m in got.items() if m is None]
    assert not missing, (missing, r.stdout)
    assert got["is_corr"].group(1) == "1"
    if "--batch" in args:
        n = int(args[args.index("--batch") + 1])
        assert re.search(r"Batch of %d queries, Is correct\?:( 1){%d}\n" % (n, n), r.stdout), r.stdout[-1500:]
        assert re.search(r"Batch of %d queries, wall \(GPU·us\): (\d+)" % n, r.stdout)
    if "--instances" in args:
        n = int(args[args.index("--instances") + 1])
        assert re.search(r"Item of %d plaintexts, Is correct\?:( 1){%d}\n" % (n, n), r.stdout), r.stdout[-1500:]
        assert re.search(r"Item of %d plaintexts \(one query, %d database instances\), device \(GPU·us\): (\d+)" % (n, n), r.stdout)
    if "--output-err" in args:
        assert "noise statistics are not produced" in r.stdout and not os.path.exists(args[args.index("--output-err") + 1])
    assert int(got["resp_sz"].group(1)) == int((2 * 2 * 2048 * (8 + 2) + 2 * 2048 * int(env.get("QPBITS", 20))) / 8)


@pytest.mark.gpu
@pytest.mark.parametrize("args,env", [
    (["6", "2", "77", "a", "--high-rate", "--seed", "6"], {}),  # the SURVEY 8c probe: ./spiral 6 2 77 a --high-rate
    (["5", "2", "9", "a", "--high-rate", "--seed", "7"], {"OUTN": "3", "TGSW": "4"}),
    (["3", "2", "5", "a", "--high-rate", "--direct-upload", "--seed", "8"], {"TEXP": "2", "TGSW": "5", "QPBITS": "19"}),
])
def test_cli_high_rate(args, env):
    e = dict(os.environ)
    e.update(env)
    r = subprocess.run([BIN] + args, capture_output=True, text=True, env=e, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    pack_re = r"\s+Packing.*:\s+([0-9]+)"  # select_params.py:393
    for k, rx in list(REGEXES.items()) + [("pack_us", pack_re)]:
        assert re.search(rx, r.stdout), (k, r.stdout)
    assert re.search(REGEXES["is_corr"], r.stdout).group(1) == "1"


# ---- spiral_amd.scheme: the JSON driver with the reference's keys (select_params.py:376-576) ----
REF_KEYS = {"exp_us", "exp_specific_us", "conv_us", "scaltomat_us", "regtogsw_us", "fdim_us", "fold_us", "pack_us", "total_us",
            "query_gen_us", "key_gen_us", "decoding_us", "resp_sz", "query_sz", "param_sz", "is_corr", "item_sz", "dbsize", "params",
            "tput", "rate", "cost"}
SMALL = {"nu_1": 4, "nu_2": 2, "p": 256, "q_prime_bits": 20, "t_GSW": 4, "t_conv": 4, "t_exp": 8, "t_exp_right": 56}


def test_scheme_analyze_and_derived_columns():
    from spiral_amd import scheme
    text = """
ScalToMat took (CPU·us): 11
RegevToGSW took (CPU·us): 22
Expansion took (CPU·us): 3.3e+02
         Total offline query size (b): 1000
          Total online query size (b): 28672
                    Response size (b): 21504
              Main expansion  (CPU·us): 330
                   Conversion (CPU·us): 33
     First dimension multiply (CPU·us): 400
                      Folding (CPU·us): 500
               Key generation (CPU·us): 1
             Query generation (CPU·us): 2
                     Decoding (CPU·us): 3
Is correct? : 1
"""
    r = scheme.analyze(text, SMALL, factor=3)
    assert r["exp_specific_us"] == 330 and r["fdim_us"] == 1200 and r["fold_us"] == 1500 and r["pack_us"] == 0
    assert r["resp_sz"] == 3 * 21504 and r["query_sz"] == 14336 and r["is_corr"] is True
    assert r["total_us"] == 330 + 33 + 1200 + 1500
    s = scheme.summarize([r, r], SMALL, item_size=20000, factor=3)
    assert set(s) >= REF_KEYS
    assert s["item_sz"] == 3 * 8192 and s["dbsize"] == 3 * 8192 * 64
    assert s["tput"] == pytest.approx(s["dbsize"] / 2700) and s["rate"] == pytest.approx(20000 / (3 * 21504))
    assert s["cost"] == pytest.approx(5.41666667e-12 * s["total_us"] + 9e-11 * s["resp_sz"])
    with pytest.raises(ValueError):
        scheme.analyze(text, dict(SMALL, n=2))  # a high-rate run must print a Packing line


def test_scheme_command_lines():
    from spiral_amd import scheme
    argv, env = scheme.command(scheme.PUBLISHED["20,256"]["spiralstream-pack"], 5, corr=False, seed=9)
    assert argv[1:] == ["10", "3", "5", "a", "--random-data", "--high-rate", "--direct-upload", "--seed", "9"]
    assert env == {"TEXP": "56", "TEXPRIGHT": "56", "TCONV": "56", "TGSW": "2", "QPBITS": "21", "PVALUE": "1024", "OUTN": "4"}
    for work in scheme.PUBLISHED.values():
        for params in work.values():
            assert {"nu_1", "nu_2", "p", "q_prime_bits", "t_GSW", "t_conv", "t_exp", "t_exp_right"} <= set(params)


@pytest.mark.gpu
@pytest.mark.parametrize("params", [SMALL, dict(SMALL, n=2, nu_1=5)])
def test_scheme_json(params):
    import json
    import sys
    r = subprocess.run([sys.executable, "-m", "spiral_amd.scheme", "--params", json.dumps(params), "--seed", "1", "--trials", "2"],
                       capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert set(out) >= REF_KEYS and out["is_corr"] == 1.0
    assert out["gpu_answer_us"] > 0

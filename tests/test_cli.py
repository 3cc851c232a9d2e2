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
    (["6", "2", "77", "a", "--random-data", "--seed", "4"], {}),
    (["5", "2", "7", "a", "--direct-upload", "--seed", "5"], {"TEXP": "2", "TGSW": "5", "QPBITS": "19"}),  # SURVEY 8c stream probe
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

import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "slow: full-size GPU case (BASELINE.json config 2)")
    config._spiral_evidence = []  # lines the full-size parity tests want in the run's tail (SHA-256s, GB/s, which branches ran)


def pytest_terminal_summary(terminalreporter, exitstatus, config):
    """`pytest -q` swallows the passing tests' prints: repeat what the full-size tests recorded at the end of the run"""
    lines = getattr(config, "_spiral_evidence", [])
    if lines:
        terminalreporter.section("full-size parity evidence")
        for ln in lines:
            terminalreporter.write_line(ln)


@pytest.fixture(scope="session")
def oracle():
    from oracle import pyoracle

    pyoracle.build()
    return pyoracle


@pytest.fixture(scope="session")
def oracle_mt(tmp_path_factory):
    """the same oracle sources built ON this machine with -march=native -fopenmp (oracle/Makefile `native`) and run on
    many threads: what the full-size parity tests use so that a 2 .. 64 GiB reference computation takes seconds.  A second
    instance of the pyoracle module bound to that library; identical results (tests/test_oracle.py checks it)."""
    import importlib.util
    import shutil
    import subprocess

    d = tmp_path_factory.mktemp("oracle_native")
    src = os.path.join(ROOT, "oracle")
    for f in ("spiral_oracle.c", "spiral_oracle_pack.c", "spiral_oracle.h", "Makefile"):
        shutil.copy(os.path.join(src, f), d)
    subprocess.check_call(["make", "-C", str(d), "-s", "native"])
    spec = importlib.util.spec_from_file_location("pyoracle_mt", os.path.join(src, "pyoracle.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.LIB_PATH = os.path.join(str(d), "liboracle_native.so")
    mod.n_threads = mod.set_threads(max(1, min(os.cpu_count() or 1, 32)))  # (scaling of this code peaks at 16-32 threads)
    return mod


@pytest.fixture
def opts():
    """opts(fold_pair=0, ...): set process-wide library options (spiral_gpu_set_option) for this test, restored afterwards"""
    import spiral_amd as sa

    saved = {}

    def set_(**kw):
        for k, v in kw.items():
            saved.setdefault(k, sa.get_option(k))
            sa.set_option(k, v)

    yield set_
    for k, v in saved.items():
        sa.set_option(k, v)

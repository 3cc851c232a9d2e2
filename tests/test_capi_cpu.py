"""CPU-side checks of the C-ABI library: it loads, exports every symbol include/spiral_gpu.h declares,
its host-only entry points agree with the reference's data, and compute entry points fail loudly
without a GPU (no CPU fallback)."""
import ctypes as C
import hashlib
import json
import os
import re
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def sa():
    import spiral_amd

    spiral_amd.build()
    return spiral_amd


def declared_symbols():
    hdr = open(os.path.join(ROOT, "include", "spiral_gpu.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(spiral_gpu_[a-z0-9_]+)\s*\(", hdr)))


def test_library_exports_every_declared_symbol(sa):
    from spiral_amd import _lib

    syms = declared_symbols()
    assert len(syms) >= 45
    raw = C.CDLL(_lib.LIB_PATH)
    for name in syms:
        assert hasattr(raw, name), f"{name} declared in include/spiral_gpu.h but not exported"
    # and the Python binding declares a prototype for each of them
    assert sorted(_lib.PROTOTYPES) == syms


def test_abi_version_and_shape(sa):
    assert sa.lib().spiral_gpu_abi_version() == 1
    s = sa.get_shape(sa.make_params(8, 7))  # config 2 of BASELINE.json, SURVEY.md section 8 header
    assert (s.dim0, s.num_per, s.m2, s.g, s.stopround, s.n_right, s.n_bits, s.qprime) == (256, 128, 24, 9, 6, 7, 312, 786433)
    s = sa.get_shape(sa.make_params(11, 9, t_gsw=4, t_conv=56, t_exp=2, qprime_bits=27, p_db=32768, direct_upload=1))
    assert (s.g, s.n_left, s.n_right, s.n_query_cts) == (0, 0, 0, 2048 + 36)
    with pytest.raises(sa.SpiralGpuError):
        sa.get_shape(sa.make_params(8, 7, qprime_bits=5))


def test_tables_match_reference_data(sa):
    gold = json.load(open(os.path.join(ROOT, "tests", "golden", "ntt_tables.json")))
    names = ["inv_p_w", "inv_p_wscaled", "inv_b_w", "inv_b_wscaled", "fwd_p_w", "fwd_p_wscaled", "fwd_b_w", "fwd_b_wscaled"]
    t = sa.get_tables()
    for r, name in enumerate(names):
        assert hashlib.sha256(np.ascontiguousarray(t[r], dtype="<u8").tobytes()).hexdigest() == gold["rows"][name]["sha256"]


def test_shapes_agree_with_oracle(sa, oracle):
    for nu1, nu2, kw in [(2, 1, {}), (4, 2, dict(t_gsw=4)), (8, 7, {}), (9, 10, dict(t_gsw=10, qprime_bits=22)),
                         (5, 2, dict(t_gsw=5, t_exp=2, qprime_bits=19, direct_upload=1))]:
        a = sa.get_shape(sa.make_params(nu1, nu2, **kw))
        b = oracle.shape_of(oracle.make_params(nu1, nu2, **kw))
        for f, _ in a._fields_:
            assert getattr(a, f) == getattr(b, f), f


def test_compute_fails_loudly_without_gpu(sa):
    if sa.lib().spiral_gpu_device_count() > 0:
        pytest.skip("a GPU is present")
    with pytest.raises(sa.SpiralGpuError):
        sa.ntt_forward(np.zeros((2, 2048), dtype=np.uint64))
    with pytest.raises(sa.SpiralGpuError):
        sa.Server(sa.make_params(2, 1))


def test_product_never_imports_the_oracle():
    """the oracle is test infrastructure: nothing under spiral_amd/ or include/ may reference it"""
    bad = []
    for base in ("spiral_amd", "include"):
        for dp, _, fs in os.walk(os.path.join(ROOT, base)):
            for f in fs:
                if f.endswith((".py", ".cpp", ".hip", ".h", ".cuh", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="ignore").read()
                    if re.search(r"liboracle|pyoracle|spiral_oracle|orc_", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad


def test_wire_form_host_functions_reject_bad_arguments():
    """the host halves of the response wire form need no device: sizes for valid parameters, 0 / an error for invalid ones"""
    import ctypes as C

    import numpy as np

    import spiral_amd as sa

    p = sa.make_params(8, 7)
    assert sa.response_wire_bytes(p) == 20480 and sa.response_wire_bytes(p, 4) == (4 * 2048 * 20 + 16 * 2048 * 10) // 8
    assert sa.response_wire_bytes(p, 0) == 0 and sa.response_wire_bytes(p, 17) == 0
    bad = sa.make_params(8, 7, qprime_bits=37)
    assert sa.response_wire_bytes(bad) == 0
    with pytest.raises(RuntimeError):
        sa.response_from_wire(bad, np.zeros(20480 + 8, dtype=np.uint8))
    out = np.zeros((3, 2, 2048), dtype=np.uint64)
    assert sa.lib().spiral_gpu_response_from_wire(C.byref(p), 2, None, out.ctypes.data_as(C.POINTER(C.c_uint64))) != 0
    assert b"null" in sa.lib().spiral_gpu_last_error()


def test_bench_self_launch_command_and_cpu_quota(monkeypatch):
    """bench.py --gpus N without a launcher: the parent builds a torch.distributed.run command for N ranks on 127.0.0.1 with the
    user's own flags and returns the children's exit code, without importing torch itself; the CPU-baseline
    thread ladder is capped by the cgroup quota"""
    import importlib
    import subprocess
    import sys

    bench = importlib.import_module("bench")
    seen = {}

    def fake_call(cmd, env=None, cwd=None):
        seen.update(cmd=cmd, env=env, cwd=cwd)
        return 7

    monkeypatch.setattr(subprocess, "call", fake_call)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    torch_loaded_before = "torch" in sys.modules
    argv = ["--gpus", "4", "--steps", "5", "--workload", "config3"]
    with pytest.raises(SystemExit) as e:
        bench.main(argv)
    assert e.value.code == 7
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"] and "--nproc-per-node=4" in cmd and "--nnodes=1" in cmd
    # the launcher's own rendezvous on a port it binds itself (no bind-close-reuse race), on the loopback address
    assert "--standalone" in cmd and cmd[cmd.index("--local-addr") + 1] == "127.0.0.1" and cmd[-len(argv):] == argv and cmd[-len(argv) - 1].endswith("bench.py")
    assert seen["env"]["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    assert ("torch" in sys.modules) == torch_loaded_before  # the parent never touches torch / the GPU
    assert 1 <= bench.cpu_quota_cores() <= (os.cpu_count() or 1)
    a = bench.parse_args(["--comm-overlap"])
    assert a.schedule == "comm-overlap" and bench.parse_args([]).schedule == "all" and bench.parse_args([]).sweep_stages == 4


def test_bench_watchdog_prints_the_line_so_far():
    """bench.py's Progress (no GPU involved): a phase that outlives the watchdog ends the process with exit code 3 and the line-so-far on stdout,
    marked partial with the phase it hung in; the same line was written to the partial file when it was recorded"""
    import json
    import subprocess
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "p.json")
        code = ("import sys, time; sys.path.insert(0, %r); import bench; p = bench.Progress(0, 1.0); "
                "p.update({'metric': 'm', 'value': 0.5, 'schedules': {'ms_per_query': {'in-order': 0.5}}}); p.arm('config2/pipelined'); time.sleep(30)") % ROOT
        r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=60, env=dict(os.environ, SPIRAL_BENCH_PARTIAL=path))
        assert r.returncode == 3
        out = json.loads(r.stdout.strip().splitlines()[-1])
        assert out["partial"] is True and out["hung_in"] == "config2/pipelined" and out["schedules"]["ms_per_query"]["in-order"] == 0.5
        assert json.load(open(path))["value"] == 0.5 and "bench.py partial: " in r.stderr
        # a disarmed watchdog never fires, and non-zero ranks never print the line
        code2 = ("import sys, time; sys.path.insert(0, %r); import bench; p = bench.Progress(1, 0.5); p.update({'value': 1}); p.arm('x'); p.disarm(); time.sleep(1.5); print('alive')") % ROOT
        r2 = subprocess.run([sys.executable, "-c", code2], capture_output=True, text=True, timeout=60, env=dict(os.environ, SPIRAL_BENCH_PARTIAL=path))
        assert r2.returncode == 0 and r2.stdout.strip() == "alive" and "partial" not in r2.stderr


def test_library_builds_from_source_on_a_clean_tree(tmp_path):
    """every object of libspiral_gpu.so compiles from source: the sources (csrc/, include/) copied WITHOUT any built artefact into a fresh directory,
    `make` there from nothing (hipcc cross-compiles gfx950 without a GPU), every object newer than the test's start, and the fresh library exports every
    declared symbol.  (The in-tree objects travel with the snapshot; this is the check that nothing depends on them.)  Also the options API -- host
    only -- and that the shipped library reads no tuning environment variable."""
    import shutil
    import subprocess
    import time

    t0 = time.time() - 1
    csrc = tmp_path / "spiral_amd" / "csrc"
    shutil.copytree(os.path.join(ROOT, "spiral_amd", "csrc"), csrc, ignore=shutil.ignore_patterns("*.o", "*.so", "*.o.*"))
    shutil.copytree(os.path.join(ROOT, "include"), tmp_path / "include")
    assert not list(csrc.glob("*.o"))
    subprocess.check_call(["make", "-C", str(csrc), "-s", "-j4", "../libspiral_gpu.so"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    objs = sorted(csrc.glob("*.o"))
    sources = sorted(list(csrc.glob("*.hip")) + list(csrc.glob("*.cpp")))
    assert len(objs) == len(sources) >= 8 and all(o.stat().st_mtime >= t0 for o in objs), "every translation unit compiled by this call"
    fresh = C.CDLL(str(tmp_path / "spiral_amd" / "libspiral_gpu.so"))
    for name in declared_symbols():
        assert hasattr(fresh, name), f"{name} missing from the from-source build"
    fresh.spiral_gpu_get_option.argtypes = [C.c_char_p, C.POINTER(C.c_int64)]
    fresh.spiral_gpu_set_option.argtypes = [C.c_char_p, C.c_int64]
    v = C.c_int64()
    for name, default in ((b"fold_pair", 1), (b"fold_chain", 1), (b"sweep_mfma_min", 2), (b"one_image", 1), (b"fwd2", -1), (b"db_stage_bytes", 64 << 20)):
        assert fresh.spiral_gpu_get_option(name, C.byref(v)) == 0 and v.value == default, name
    assert fresh.spiral_gpu_set_option(b"fwd2", 1) == 0 and fresh.spiral_gpu_get_option(b"fwd2", C.byref(v)) == 0 and v.value == 1
    assert fresh.spiral_gpu_set_option(b"no_such_option", 1) != 0 and fresh.spiral_gpu_set_option(b"fwd2", 7) != 0
    # the shipped sources read exactly three environment variables (README.md); the tuning thresholds exist only behind -DSPIRAL_TUNING
    env_reads = set()
    for f in sources + list(csrc.glob("*.h")):
        env_reads |= set(re.findall(r'\bgetenv\("(SPIRAL_[A-Z0-9_]+)"\)', f.read_text()))
    assert env_reads == {"SPIRAL_FOLD_PAIR", "SPIRAL_SWEEP_MFMA", "SPIRAL_DB_STAGE_BYTES"}, env_reads


def test_graft_entry_build_reports_its_mode(capsys):
    sys.path.insert(0, ROOT)
    import __graft_entry__ as g

    g.build()
    out = capsys.readouterr().out
    assert "build_mode:" in out

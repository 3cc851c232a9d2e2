"""The oracle's C restatement under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build, the only place sanitizers run on this
pool): whole base and SpiralPack answers, both query forms, odd gadget dimensions, the wire form -- no report allowed.  The parity
claims rest on this code; an out-of-bounds read that happens to produce the reference's numbers would void them."""
import os
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

DRIVER = r'''
import sys
sys.path.insert(0, {root!r})
from oracle import pyoracle as O
O.LIB_PATH = {lib!r}
for kw in (dict(t_gsw=4), dict(t_gsw=5, t_exp=2, qprime_bits=19, direct_upload=1), dict(t_gsw=3, t_conv=7, t_exp=5, t_exp_right=28, p_db=4)):
    po = O.make_params(4, 2, **kw)
    cl = O.Client(po, seed=1)
    pp = cl.pub_params()
    db = O.gen_db(po, 3)
    for idx in (0, 63):
        q = cl.query(idx)
        fin = O.answer(po, q, *pp, db)
po = O.make_params(4, 2, t_gsw=4)
for out_n in (2, 3):
    db = O.pack_gen_db(po, out_n, 5)
    cl = O.PackClient(po, out_n, seed=2)
    pp = cl.pub_params()
    resp, packed = O.pack_answer(po, out_n, cl.query(17), *pp, db)
    assert (cl.decode(resp) == O.pack_db_item(po, out_n, 5, 17)).all()
    w = O.response_to_wire(po, resp, out_n)
    assert (O.response_from_wire(po, w, out_n) == resp).all()
print("SANITIZED-RUN-COMPLETE")
'''


def test_oracle_is_clean_under_asan_and_ubsan(tmp_path):
    gcc = shutil.which("gcc")
    if not gcc:
        pytest.skip("no gcc")
    libasan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    if not os.path.isabs(libasan) or not os.path.exists(libasan):
        pytest.skip("no libasan in this toolchain")
    lib = str(tmp_path / "liboracle_san.so")
    src = [os.path.join(ROOT, "oracle", f) for f in ("spiral_oracle.c", "spiral_oracle_pack.c")]
    subprocess.check_call([gcc, "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-fPIC", "-shared",
                           "-I", os.path.join(ROOT, "oracle"), "-o", lib] + src + ["-lm"])
    env = dict(os.environ, LD_PRELOAD=libasan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([sys.executable, "-c", DRIVER.format(root=ROOT, lib=lib)], capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0 and "SANITIZED-RUN-COMPLETE" in r.stdout, (r.stdout[-1500:], r.stderr[-3000:])
    assert "runtime error" not in r.stderr and "AddressSanitizer" not in r.stderr, r.stderr[-3000:]

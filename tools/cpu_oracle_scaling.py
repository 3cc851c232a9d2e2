import sys, time, os
"""per-stage time of the oracle's threaded native build on this machine: tools/cpu_oracle_scaling.py 1 16 64 128 ..."""
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import shutil, subprocess, tempfile
import numpy as np
from oracle import pyoracle as O
d = tempfile.mkdtemp(prefix="onat_")
for f in ("spiral_oracle.c", "spiral_oracle_pack.c", "spiral_oracle.h", "Makefile"):
    shutil.copy(os.path.join(ROOT, "oracle", f), d)
subprocess.check_call(["make", "-C", d, "-s", "native"])
O.LIB_PATH = os.path.join(d, "liboracle_native.so")
po = O.make_params(8, 7); s = O.shape_of(po)
rng = np.random.default_rng(7)
db = O.fill_db_random(99, O.db_words(po))
mk = lambda shape: np.ascontiguousarray(np.stack([rng.integers(0, m, size=shape + (O.N,), dtype=np.uint64) for m in (O.P, O.B)], axis=-2))
wl, wr = mk((s.n_left, 2, po.t_exp)), mk((s.n_right, 2, po.t_exp_right))
w, v = mk((3, 8)), mk((3, 8)); q = mk((1, 2))
O.set_ntt_simd(True)  # the reference's USE_AVX2 transforms, as bench.py's cpu_baseline
print("transforms:", O.ntt_isa(), "| sweep cell:", O.sweep_isa(), "| OMP env:", {k: v for k, v in os.environ.items() if k.startswith(("OMP_", "GOMP_"))})
for th in [int(x) for x in sys.argv[1:]]:
    O.set_threads(th)
    for rep in range(2):  # the second pass is the warm one (scratch blocks cached by malloc, threads started)
        t0 = time.perf_counter(); cv = O.stage_expand(po, q, wl, wr); t1 = time.perf_counter()
        cts, gsw = O.stage_convert(po, cv, w, v); t2 = time.perf_counter()
        raw = O.stage_first_dim(po, cts, db); t3 = time.perf_counter()
        fin = O.stage_fold(po, raw, gsw); t4 = time.perf_counter()
        re = O.reorient_ciphertexts(cts); t5 = time.perf_counter()
        O.multiply_query_by_database(re, db, s.dim0, s.num_per); t6 = time.perf_counter()
    print(th, "threads: expand %.0f convert %.0f first_dim %.0f (sweep loop alone %.0f = %.1f GB/s of NTT-form database) fold %.0f total %.0f ms" % (
        (t1-t0)*1e3, (t2-t1)*1e3, (t3-t2)*1e3, (t6-t5)*1e3, O.db_words(po) * 8 / (t6-t5) / 1e9, (t4-t3)*1e3, (t4-t0)*1e3), flush=True)

"""the batched first-dimension sweep alone (spiral_gpu_server_first_dim_batch): wall us per launch for B = 1 .. 4 queries per pass at config 2
(or --nu1/--nu2), `reps` back-to-back launches, best of 5.  SPIRAL_LIB=<path> selects another build (A/B)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spiral_amd as sa

opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
nu1, nu2, reps = int(opts.get("nu1", 8)), int(opts.get("nu2", 7)), int(opts.get("reps", 30))
pg = sa.make_params(nu1, nu2)
s = sa.get_shape(pg)
rng = np.random.default_rng(1)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
owner = sa.Server(pg)
owner.fill_db_random(3)
lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(3)]
for ln in lanes:
    ln.set_pub_params(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 2 * pg.t_conv)), mk((3, 2 * pg.t_conv)))
    ln.set_query(mk((s.n_query_cts, 2)))
    ln.run_pre()
    ln.sync()
out = []
for B in (1, 2, 3, 4):
    group = lanes[:B]
    for _ in range(5):
        sa.first_dim_batch(group)
    for ln in group: ln.sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps):
            sa.first_dim_batch(group)
        for ln in group: ln.sync()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    out.append(f"B={B}: {best:7.1f} us")
print(f"{os.path.basename(os.environ.get('SPIRAL_LIB', 'product')):28s} sweep per launch: " + "  ".join(out), flush=True)

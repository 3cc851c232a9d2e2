"""single-query latency on the two forms of the one database image, alternating on one server: the vector-ALU sweep on the packed form against sweep_mfma_kernel<1> on the limb planes
(whole-query hipGraph replays, wall us per query).  usage: python tools/image_form_ab.py [--nu1=8 --nu2=7 --reps=60 --rounds=5]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import spiral_amd as sa
from spiral_amd import server as SV

opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
nu1, nu2, reps, rounds = int(opts.get("nu1", 8)), int(opts.get("nu2", 7)), int(opts.get("reps", 60)), int(opts.get("rounds", 5))
kw = {k: int(opts[k]) for k in ("t_gsw", "t_conv", "t_exp", "t_exp_right", "qprime_bits") if k in opts}
pg = sa.make_params(nu1, nu2, **kw)
s = sa.get_shape(pg)
rng = np.random.default_rng(1)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
srv = sa.Server(pg)
srv.fill_db_random(3)
st = torch.cuda.Stream()
srv.set_stream(st.cuda_stream)
srv.set_pub_params(mk((max(s.n_left, 1), 2, pg.t_exp)), mk((max(s.n_right, 1), 2, pg.t_exp_right)), mk((3, 2 * pg.t_conv)), mk((3, 2 * pg.t_conv)))
srv.set_query(mk((s.n_query_cts, 2)))
srv.use_graphs(True)
for _ in range(40):
    srv.run_query()
srv.sync()
for rnd in range(rounds):
    for fmt, name in ((SV.DB_PACKED, "packed, sweep_kernel"), (SV.DB_LIMBS, "limb planes, sweep_mfma_kernel<1>")):
        srv.set_db_format(fmt)
        for _ in range(10):
            srv.run_query()
        srv.sync()
        t0 = time.perf_counter()
        for _ in range(reps):
            srv.run_query()
        srv.sync()
        us = (time.perf_counter() - t0) / reps * 1e6
        print(f"round {rnd}: {name:36s} {us:8.1f} us per query   sweep launch {srv.time_sweep(8) * 1e3:7.1f} us", flush=True)

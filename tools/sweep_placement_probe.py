"""Does the sweep's launch time depend on where the database image landed in HBM?  N servers alive at once (each its own image, same contents),
time_sweep on each, several rounds; prints the device address of every accumulator buffer (a proxy for the allocation order) and the times."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spiral_amd as sa
pg = sa.make_params(8, 7); s = sa.get_shape(pg)
rng = np.random.default_rng(1)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
pp = (mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 8)), mk((3, 8)))
q = mk((1, 2))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 6
srvs = []
for i in range(n):
    srv = sa.Server(pg); srv.fill_db_random(3); srv.set_pub_params(*pp); srv.set_query(q); srv.run_pre(); srv.sync()
    srvs.append(srv)
for rnd in range(3):
    print("round", rnd, " ".join(f"{srv.time_sweep(12) * 1e3:6.1f}" for srv in srvs), flush=True)
print("acc addresses:", " ".join(hex(srv.acc()[0]) for srv in srvs))
for srv in srvs: srv.close()
# and sequential create / time / close, as tools/stage_ab.py does
for i in range(6):
    srv = sa.Server(pg); srv.fill_db_random(3); srv.set_pub_params(*pp); srv.set_query(q); srv.run_pre(); srv.sync()
    print("sequential", i, f"{srv.time_sweep(12) * 1e3:6.1f}", hex(srv.acc()[0]), flush=True)
    srv.close()

"""stage times of answer_resident() with an alternative build of the library (SPIRAL_LIB=path)"""
import os, sys; sys.path.insert(0, '.')
import numpy as np
import spiral_amd as sa
pg = sa.make_params(8, 7); s = sa.get_shape(pg)
srv = sa.Server(pg); srv.fill_db_random(3)
rng = np.random.default_rng(1)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
srv.set_pub_params(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 8)), mk((3, 8)))
srv.set_query(mk((1, 2)))
us = [srv.answer_resident() for _ in range(12)][2:]
print(os.environ.get("SPIRAL_LIB", "product"), {k: round(float(np.median([u[k] for u in us])), 1) for k in us[0]})

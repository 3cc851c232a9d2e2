import sys; sys.path.insert(0,'.')
import numpy as np, spiral_amd as sa
pg=sa.make_params(8,7); s=sa.get_shape(pg)
srv=sa.Server(pg); srv.gen_db(1234)
rng=np.random.default_rng(1)
mk=lambda shape: np.stack([rng.integers(0,m,size=shape+(sa.N,),dtype=np.uint64) for m in (sa.P,sa.B)],axis=-2)
srv.set_pub_params(mk((s.n_left,2,pg.t_exp)),mk((s.n_right,2,pg.t_exp_right)),mk((3,8)),mk((3,8)))
srv.set_query(mk((1,2)))
srv.answer_resident()
for it in (1,5,20,50):
    print(it, round(srv.time_sweep(it)*1e3,1),'us')
print(srv.answer_resident())
srv.fill_db_random(5)
for it in (5,20):
    print('randdb',it, round(srv.time_sweep(it)*1e3,1),'us')

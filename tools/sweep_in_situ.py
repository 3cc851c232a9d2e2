"""Where does the in-pipeline sweep lose ~10% vs back-to-back sweeps?  A/B in one process with HIP events."""
import sys; sys.path.insert(0, '.')
import numpy as np, torch, spiral_amd as sa
pg = sa.make_params(8, 7); s = sa.get_shape(pg)
dev = torch.device('cuda', 0)
srv = sa.Server(pg); srv.gen_db(1234)
rng = np.random.default_rng(1)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
srv.set_pub_params(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 8)), mk((3, 8)))
srv.set_query(mk((1, 2)))
st = torch.cuda.Stream(device=dev); srv.set_stream(st.cuda_stream)
srv.answer_resident()
def run(label, before, n=12):
    ts = []
    with torch.cuda.stream(st):
        for _ in range(n):
            before()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st); srv.first_dim(); e1.record(st)
            ts.append((e0, e1))
        torch.cuda.synchronize()
    v = sorted(a.elapsed_time(b) * 1e3 for a, b in ts[2:])
    print(f"{label:34s} median {v[len(v)//2]:7.1f}  min {v[0]:7.1f}")
big = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for rep in range(2):
    run("back-to-back", lambda: None)
    run("after expand+convert", lambda: srv.run_pre())
    run("after lift+fold+finish", lambda: srv.run_post())
    run("after expand only", lambda: srv.expand())
    run("after convert only", lambda: srv.convert())
    run("after 512MB memset", lambda: big.zero_())
    run("after pre + sync", lambda: (srv.run_pre(), torch.cuda.synchronize()))

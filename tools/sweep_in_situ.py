"""why is the sweep slower inside bench.py than standalone? A/B the candidate causes in one process."""
import sys, time; sys.path.insert(0, '.')
import numpy as np, torch, spiral_amd as sa
pg = sa.make_params(8, 7); s = sa.get_shape(pg)
dev = torch.device('cuda', 0)
srv = sa.Server(pg); srv.gen_db(1234)
rng = np.random.default_rng(1)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
srv.set_pub_params(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 8)), mk((3, 8)))
srv.set_query(mk((1, 2)))
def loop(n, label):
    us = []
    for _ in range(n):
        us.append(srv.answer_resident())
    print(f"{label:40s} sweep {np.median([u['sweep_kernel_us'] for u in us]):7.1f}  total {np.median([u['total_us'] for u in us]):7.1f}  standalone {srv.time_sweep(10)*1e3:7.1f}")
loop(3, "own stream, own acc, few")
loop(30, "own stream, own acc, 30 back-to-back")
acc = torch.zeros(s.num_per * 6 * sa.N, dtype=torch.int64, device=dev)
srv.set_acc(acc.data_ptr()); loop(30, "own stream, torch acc")
st = torch.cuda.Stream(device=dev); srv.set_stream(st.cuda_stream); loop(30, "torch stream, torch acc")
srv.use_graphs(True); loop(30, "torch stream, torch acc, graphs")
# bench-style: no sync between steps
with torch.cuda.stream(st):
    ev = [[torch.cuda.Event(enable_timing=True) for _ in range(3)] for _ in range(30)]
    for k in range(30):
        srv.run_pre(); ev[k][0].record(st); srv.first_dim(); ev[k][1].record(st); srv.run_post(); ev[k][2].record(st)
    torch.cuda.synchronize()
    print("bench-style async: sweep", np.median([e[0].elapsed_time(e[1]) * 1e3 for e in ev]), " post", np.median([e[1].elapsed_time(e[2]) * 1e3 for e in ev]))

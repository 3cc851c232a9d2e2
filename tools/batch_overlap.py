"""G batches of B whole queries in flight on G streams (run_query_batch per group, groups alternating): does one batch's HBM-bound sweep hide under another's
VALU-bound expansion / folding?  usage: tools/batch_overlap.py [--groups=2] [--lanes=4] [--reps=20] [--nu1=8 --nu2=7] [--offset_us=0]
--offset_us: the second group's first batch is enqueued that much later than the first's (then both queues stay full): do the two streams run better out of phase?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import spiral_amd as sa

opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
nu1, nu2, reps = int(opts.get("nu1", 8)), int(opts.get("nu2", 7)), int(opts.get("reps", 20))
G, B, offset_us = int(opts.get("groups", 2)), int(opts.get("lanes", 4)), float(opts.get("offset_us", 0))
pg = sa.make_params(nu1, nu2)
s = sa.get_shape(pg)
rng = np.random.default_rng(1)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
owner = sa.Server(pg)
owner.fill_db_random(3)
lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(G * B - 1)]
streams = [torch.cuda.Stream() for _ in range(G)]
groups = [lanes[g * B:(g + 1) * B] for g in range(G)]
for g, grp in enumerate(groups):
    for ln in grp:
        ln.set_stream(streams[g].cuda_stream)
        ln.set_pub_params(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 2 * pg.t_conv)), mk((3, 2 * pg.t_conv)))
        ln.set_query(mk((s.n_query_cts, 2)))
        ln.use_graphs(True)
for n_g in range(1, G + 1):
    act = groups[:n_g]
    for _ in range(4):
        for grp in act: sa.run_query_batch(grp)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        if offset_us and n_g > 1:  # group 0 gets a head start of offset_us (and one batch more than the others in this pass)
            sa.run_query_batch(act[0])
            while (time.perf_counter() - t0) * 1e6 < offset_us: pass
        for _ in range(reps):
            for grp in act: sa.run_query_batch(grp)
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / (reps + (1.0 / n_g if offset_us and n_g > 1 else 0.0)) * 1e6)
    print(f"{n_g} x {B} in flight: {best:8.1f} us per round of {n_g * B} queries, {n_g * B * 1e6 / best:7.1f} queries/s", flush=True)

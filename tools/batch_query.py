"""B whole queries per launch sequence (spiral_gpu_server_run_query_batch) at config 2 (or --nu1/--nu2): wall us per batch over hipGraph replays.
usage: tools/batch_query.py [B ...] [--reps=40] [--opt:NAME=VALUE ...] [--nu1=8 --nu2=7] [--streams=1]      (under rocprofv3 --kernel-trace for the timeline: tools/trace_summary.py)
--streams=1: every lane on its own stream, as a serving loop would hold them (the batch call orders them around its launch sequence with events)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch  # (before the library initialises the device)
import spiral_amd as sa

Bs = [int(a) for a in sys.argv[1:] if not a.startswith("--")] or [1, 2, 4, 8]
opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--") and not a.startswith("--opt:"))
for a in sys.argv[1:]:
    if a.startswith("--opt:"):  # library options (spiral_gpu_set_option) before any server exists, e.g. --opt:fwd2_min=4096
        k, v = a[6:].split("=")
        sa.set_option(k, int(v))
nu1, nu2, reps = int(opts.get("nu1", 8)), int(opts.get("nu2", 7)), int(opts.get("reps", 40))
kw = {k: int(opts[k]) for k in ("t_gsw", "t_conv", "t_exp", "t_exp_right") if k in opts}
pg = sa.make_params(nu1, nu2, **kw)
s = sa.get_shape(pg)
rng = np.random.default_rng(1)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
owner = sa.Server(pg)
owner.fill_db_random(3)
lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(max(Bs) - 1)]
for ln in lanes:  # every lane its own keys and query
    ln.set_pub_params(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 2 * pg.t_conv)), mk((3, 2 * pg.t_conv)))
    ln.set_query(mk((s.n_query_cts, 2)))
    ln.use_graphs(True)
# one stream for all lanes (not the legacy default stream, whose implicit synchronisation costs a batch 5-10 %), or one per lane
streams = [torch.cuda.Stream() for _ in lanes] if int(opts.get("streams", 0)) else [torch.cuda.Stream()] * len(lanes)
for ln, st in zip(lanes, streams):
    ln.set_stream(st.cuda_stream)
for B in Bs:
    group = lanes[:B]
    for _ in range(5):
        sa.run_query_batch(group)
    owner.sync()
    best = 1e9
    for _ in range(3):
        t0 = time.perf_counter()
        for _ in range(reps):
            sa.run_query_batch(group)
        owner.sync()
        best = min(best, (time.perf_counter() - t0) / reps * 1e6)
    print(f"B={B}: {best:8.1f} us per batch, {best / B:7.1f} us per query, {B * 1e6 / best:7.1f} queries/s", flush=True)

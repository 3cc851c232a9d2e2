"""the batched sweep on the matrix cores (sweep_mfma.hip) against the single-query sweep on the same random database and random query
records: accumulators of every lane must be bit-identical; then wall us per launch for B = 1 .. 8 queries per pass (back-to-back launches on one stream).
  python tools/sweep_mfma_check.py [--nu1=8 --nu2=7 --reps=30 --min=1]      (--min: SPIRAL_SWEEP_MFMA threshold, 1 = every batch size)"""
import os, sys, time
opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
os.environ["SPIRAL_SWEEP_MFMA"] = opts.get("min", "1")
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import spiral_amd as sa
from spiral_amd import server as SV

nu1, nu2, reps, nl = int(opts.get("nu1", 8)), int(opts.get("nu2", 7)), int(opts.get("reps", 30)), int(opts.get("lanes", 8))
pg = sa.make_params(nu1, nu2)
s = sa.get_shape(pg)
rng = np.random.default_rng(1)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
owner = sa.Server(pg)
owner.fill_db_random(3)
lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(nl - 1)]
want, accs = [], []
acc_words = s.num_per * 6 * sa.N  # PK words: [num_per][3][2][N]
stream = torch.cuda.Stream()  # one stream for all lanes: first_dim_batch then needs no event ordering between them
for ln in lanes:
    ln.set_stream(stream.cuda_stream)
    accs.append(torch.zeros(acc_words, dtype=torch.int64, device="cuda"))
    ln.set_acc(accs[-1].data_ptr())
    ln.set_pub_params(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 2 * pg.t_conv)), mk((3, 2 * pg.t_conv)))
    ln.set_query(mk((s.n_query_cts, 2)))
    ln.run_pre()
    ln.first_dim()
    ln.sync()
    torch.cuda.synchronize()
    want.append(accs[-1].clone())
bad = 0
for B in range(2, nl + 1):
    group = lanes[:B]
    for a in accs:
        a.fill_(-1)
    torch.cuda.synchronize()
    sa.first_dim_batch(group)
    for ln in group: ln.sync()
    torch.cuda.synchronize()
    for b, ln in enumerate(group):
        got = accs[b]
        if not torch.equal(got, want[b]):
            d = torch.nonzero(got != want[b]).flatten()
            i = int(d[0])
            print(f"B={B} lane {b}: {len(d)} of {got.numel()} words differ, first at word {i} (poly {i // sa.N}, z {i % sa.N}): got {int(got[i]) & (2**64 - 1):#x} want {int(want[b][i]) & (2**64 - 1):#x}", flush=True)
            bad += 1
print("accumulators identical for B = 2 .. %d" % nl if not bad else f"MISMATCH in {bad} (B, lane) pairs", flush=True)
out = []
for B in range(1, nl + 1):
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
    out.append(f"B={B}: {best:6.1f}")
print(f"nu1={nu1} nu2={nu2} sweep us per launch: " + "  ".join(out), flush=True)
sys.exit(1 if bad else 0)

"""Repeatability soak: the same query replayed N times (whole-query hipGraph; then eight-lane batches on two streams in flight), every replay's folded ciphertext + response + accumulators
hashed and compared with the first.  A missing barrier or an unordered exchange between launches shows up as a rare mismatch long before it shows up in a parity test.
usage: tools/determinism_soak.py [--queries=20000] [--batches=3000] [--nu1=8 --nu2=7]"""
import hashlib, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import spiral_amd as sa
from spiral_amd import server as SV

opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
nq, nb = int(opts.get("queries", 20000)), int(opts.get("batches", 3000))
pg = sa.make_params(int(opts.get("nu1", 8)), int(opts.get("nu2", 7)))
s = sa.get_shape(pg)
rng = np.random.default_rng(5)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
owner = sa.Server(pg)
owner.gen_db(77)
lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(15)]
for ln in lanes:
    ln.set_pub_params(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 2 * pg.t_conv)), mk((3, 2 * pg.t_conv)))
    ln.set_query(mk((s.n_query_cts, 2)))
    ln.use_graphs(True)


def digest(ln, acc=True):
    h = hashlib.sha256()
    for which in ((SV.BUF_ACC,) if acc else ()) + (SV.BUF_FINAL, SV.BUF_RESPONSE):
        h.update(ln.read(which).tobytes())
    return h.hexdigest()


# (a) one query, replayed; checked every `every` replays (a read-back synchronises, so most replays run back to back as in the benchmark)
owner.run_query()
ref, ref_out = digest(owner), digest(owner, acc=False)
every, bad, t0 = 250, 0, time.time()
for k in range(nq):
    owner.run_query()
    if k % 2 == 0:  # every second replay is read back (the other one runs straight behind its predecessor, as in the benchmark)
        bad += digest(owner, acc=False) != ref_out
    if (k + 1) % every == 0:
        bad += digest(owner) != ref
print(f"one query: {nq} replays; folded ciphertext + response of every second replay and the 24 MiB of accumulators of every {every}th compared with the first: {bad} mismatches "
      f"({time.time() - t0:.0f} s); sha256 {ref[:16]}", flush=True)

# (b) two batches of eight in flight on two streams, replayed; every lane checked against its own first result
st = [torch.cuda.Stream(), torch.cuda.Stream()]
groups = [lanes[:8], lanes[8:]]
for g, stream in zip(groups, st):
    for ln in g:
        ln.set_stream(stream.cuda_stream)
for g in groups:
    sa.run_query_batch(g)
for g in groups:
    g[0].sync()
refs = [[digest(ln) for ln in g] for g in groups]
refs_out = [[digest(ln, acc=False) for ln in g] for g in groups]
single = []
for ln in lanes[:2]:  # the batch's lane results are what the lane computes alone
    ln.run_query()
    single.append(digest(ln))
assert single == refs[0][:2], "a batch lane differs from the same query run alone"
every, bad, t0 = 100, 0, time.time()
for k in range(nb):
    for g in groups:
        sa.run_query_batch(g)
    if k % 4 == 0:
        for g, r in zip(groups, refs_out):
            g[0].sync()
            bad += sum(digest(ln, acc=False) != x for ln, x in zip(g, r))
    if (k + 1) % every == 0:
        for g, r in zip(groups, refs):
            g[0].sync()
            bad += sum(digest(ln) != x for ln, x in zip(g, r))
print(f"two eight-query batches in flight: {nb} rounds ({16 * nb} queries); all 16 lanes' folded ciphertexts + responses every fourth round and their accumulators every {every}th compared with "
      f"their first results (lanes 0, 1 also with the same query run alone): {bad} mismatches ({time.time() - t0:.0f} s)", flush=True)
sys.exit(1 if bad else 0)

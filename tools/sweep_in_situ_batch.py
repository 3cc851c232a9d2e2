"""The eight-query matrix-core sweep takes ~0.40 ms back to back and 0.45-0.49 ms inside a batch: clocks (it is issue-bound, and follows 1.3 ms of VALU-bound launches)
or the memory system (it follows 400 MB of record / key writes)?  time_sweep_batch right after different predecessors, HIP events on the launch stream.
usage: python tools/sweep_in_situ_batch.py [B=8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import spiral_amd as sa

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
pg = sa.make_params(8, 7)
s = sa.get_shape(pg)
rng = np.random.default_rng(1)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
owner = sa.Server(pg)
owner.fill_db_random(3)
lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(B - 1)]
st = torch.cuda.Stream()
for ln in lanes:
    ln.set_pub_params(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 2 * pg.t_conv)), mk((3, 2 * pg.t_conv)))
    ln.set_query(mk((s.n_query_cts, 2)))
    ln.use_graphs(True)
    ln.set_stream(st.cuda_stream)
for _ in range(3):
    sa.run_query_batch(lanes)
owner.sync()
big = torch.empty(1 << 30, dtype=torch.uint8, device="cuda")


def measure(label, before, n=8):
    v = []
    for _ in range(n):
        before()
        v.append(sa.time_sweep_batch(lanes, 1) * 1e3)  # (synchronises the device first: what is measured is the launch right after `before` has drained)
    v.sort()
    print(f"{label:46s} median {v[len(v) // 2]:7.1f} us  min {v[0]:7.1f}  max {v[-1]:7.1f}", flush=True)


def busy(ms):  # VALU-bound work: the transforms of a batch, no sweep
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        sa.time_ntt(16384, 2)


for rep in range(2):
    measure("after 50 ms idle", lambda: time.sleep(0.05))
    measure("back to back (12 launches averaged)", lambda: None, n=1) if False else print(f"{'back to back, 12 launches':46s} avg    {sa.time_sweep_batch(lanes, 12) * 1e3:7.1f} us", flush=True)
    measure("after 3 whole batches (hot)", lambda: [sa.run_query_batch(lanes) for _ in range(3)])
    measure("after 20 ms of transform launches", lambda: busy(20))
    measure("after a 1 GiB memset (dirty caches)", lambda: (big.zero_(), torch.cuda.synchronize()))
    with torch.cuda.stream(st):
        measure("after 3 batches + 1 GiB memset", lambda: ([sa.run_query_batch(lanes) for _ in range(3)], big.zero_()))

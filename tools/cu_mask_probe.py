"""Two eight-query batches in flight on two streams overlap only at kernel boundaries: a full-grid kernel fills every CU's wave slots, so one batch's HBM-bound sweep does not run UNDER
the other's VALU-bound transforms.  Does partitioning the chip help -- each batch on its own half of the CUs (hipExtStreamCreateWithCUMask), so that one half streams the database while the
other half transforms?  usage: python tools/cu_mask_probe.py [--lanes=8] [--reps=20]"""
import ctypes as C
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import spiral_amd as sa

opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
B, reps = int(opts.get("lanes", 8)), int(opts.get("reps", 20))
sa.lib()
hip = C.CDLL("libamdhip64.so")


def stream(words):
    st = C.c_void_p()
    arr = (C.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(st), len(words), arr)
    assert rc == 0, rc
    return st.value


FULL = [0xFFFFFFFF] * 8
MASKS = {
    "all 256 CUs": (FULL, FULL),
    "words 0-3 / words 4-7": ([0xFFFFFFFF] * 4 + [0] * 4, [0] * 4 + [0xFFFFFFFF] * 4),
    "even bits / odd bits": ([0x55555555] * 8, [0xAAAAAAAA] * 8),
    "low / high half of every word": ([0x0000FFFF] * 8, [0xFFFF0000] * 8),
    "3/4 : 1/4 (low 24 / high 8 bits of every word)": ([0x00FFFFFF] * 8, [0xFF000000] * 8),
}
pg = sa.make_params(8, 7)
s = sa.get_shape(pg)
rng = np.random.default_rng(1)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
owner = sa.Server(pg)
owner.fill_db_random(3)
lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(2 * B - 1)]
for ln in lanes:
    ln.set_pub_params(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 2 * pg.t_conv)), mk((3, 2 * pg.t_conv)))
    ln.set_query(mk((s.n_query_cts, 2)))
    ln.use_graphs(True)
groups = [lanes[:B], lanes[B:]]


def sync_all():
    for g in groups:
        g[0].sync()


for name, (m0, m1) in MASKS.items():
    sts = [stream(m0), stream(m1)]
    for g, st in zip(groups, sts):
        for ln in g:
            ln.set_stream(st)
    out = []
    for act in ([groups[0]], groups):
        for _ in range(3):
            for g in act: sa.run_query_batch(g)
        sync_all()
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter()
            for _ in range(reps):
                for g in act: sa.run_query_batch(g)
            sync_all()
            best = min(best, (time.perf_counter() - t0) / reps * 1e6)
        out.append(f"{len(act)} x {B}: {best:8.1f} us per round = {len(act) * B * 1e6 / best:7.1f} queries/s")
    # one HBM-bound and one VALU-bound piece alone on the first stream's CUs: the sweep of one query and its expansion + conversion
    owner.set_db_format(0)
    owner.run_pre(); owner.sync()
    sw = owner.time_sweep(6) * 1e3
    t0 = time.perf_counter()
    for _ in range(10): owner.run_pre()
    owner.sync()
    pre = (time.perf_counter() - t0) / 10 * 1e6
    print(f"{name:48s} {out[0]} | {out[1]} | one-query sweep {sw:6.1f} us, expand + convert {pre:6.1f} us on stream 0's CUs", flush=True)

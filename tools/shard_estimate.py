"""what ONE rank of a G-GPU answer computes, timed on one GPU (no collectives): tools/shard_estimate.py [G ...]
sharded expansion + pack, unpack + convert + sweep of a 1/G database shard, local fold rounds, root fold rounds."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import spiral_amd as sa

def timed(fn, stream, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(n): fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n

pg = sa.make_params(8, 7); shp = sa.get_shape(pg)
rng = np.random.default_rng(1)
mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
pp = (mk((shp.n_left, 2, pg.t_exp)), mk((shp.n_right, 2, pg.t_exp_right)), mk((3, 8)), mk((3, 8)))
q = mk((1, 2))
dev = torch.device("cuda", 0)
for G in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    per = shp.dim0 // G
    srv = sa.Server(pg, 0, 0, per)
    stream = torch.cuda.Stream(device=dev); srv.set_stream(stream.cuda_stream)
    srv.fill_db_random(3); srv.set_pub_params(*pp); srv.set_query(q)
    acc = torch.zeros(shp.num_per * 6 * sa.N, dtype=torch.int64, device=dev); srv.set_acc(acc.data_ptr())
    srv.set_fold_ranks(G)
    chunk = torch.zeros(acc.numel() // G, dtype=torch.int64, device=dev)
    ct = torch.zeros(6 * sa.N, dtype=torch.int64, device=dev); gathered = torch.zeros(G * 6 * sa.N, dtype=torch.int64, device=dev)
    with torch.cuda.stream(stream):
        srv.use_graphs(True)
        t_rep = timed(srv.run_pre_sweep, stream)
        res = {"replicated expand+convert+sweep": t_rep}
        if G > 1:
            srv.set_expand_shard(0, G)
            bits = torch.zeros(srv.gsw_bits_words(), dtype=torch.int64, device=dev); bits_all = torch.zeros(G * bits.numel(), dtype=torch.int64, device=dev)
            res["sharded expand+pack"] = timed(lambda: srv.run_expand_pack(bits.data_ptr()), stream)
            res["unpack+convert+sweep"] = timed(lambda: srv.run_unpack_convert_sweep(bits_all.data_ptr()), stream)
        res["fold_local"] = timed(lambda: srv.fold_local(chunk.data_ptr(), ct.data_ptr()), stream)
        res["fold_root"] = timed(lambda: srv.fold_root(gathered.data_ptr()), stream)
    print(f"G={G}: " + ", ".join(f"{k} {v:.0f} us" for k, v in res.items()))
    srv.close()

"""CPU coverage of the N > 1 path (gloo, world size 2): the j-shard partition and the single packed
sum-reduce of spiral_amd/dist.py, checked with the oracle's sweep standing in for the HIP kernel
(the GPU-side equivalence of sharded and unsharded accumulators is tests/test_gpu_parity.py::
test_sharded_first_dim_sums_to_unsharded)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from oracle import pyoracle as O
    from spiral_amd import dist as sdist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    po = O.make_params(3, 2, t_gsw=4)
    s = O.shape_of(po)
    rng = np.random.default_rng(123)  # same inputs on both ranks
    cts = np.stack([rng.integers(0, m, size=(s.dim0, 3, 2, O.N), dtype=np.uint64) for m in (O.P, O.B)], axis=3)
    db = O.gen_db(po, 9)
    j0, j1 = sdist.shard_range(rank, world, s.dim0)
    # this rank's shard of the reference-layout database and of the expanded query
    dbv = db.reshape(O.N, s.num_per, 2, s.dim0, 2)[:, :, :, j0:j1, :]
    part = O.multiply_query_by_database(O.reorient_ciphertexts(cts[j0:j1]), np.ascontiguousarray(dbv).reshape(-1), j1 - j0, s.num_per)
    packed = (part[..., 0, :] | (part[..., 1, :] << np.uint64(32))).astype(np.uint64)  # the sweep kernel's output words
    acc = torch.from_numpy(packed.view(np.int64).copy())
    sdist.reduce_accumulators(acc, dst=0)
    if rank == 0:
        tot = acc.numpy().view(np.uint64)
        lo, hi = (tot & np.uint64(0xFFFFFFFF)) % np.uint64(O.P), (tot >> np.uint64(32)) % np.uint64(O.B)  # lift(reduce_first=True)
        full = O.multiply_query_by_database(O.reorient_ciphertexts(cts), db, s.dim0, s.num_per)
        q.put(bool((lo == full[..., 0, :]).all() and (hi == full[..., 1, :]).all()))
    dist.destroy_process_group()


def test_two_rank_shard_and_reduce_equals_unsharded():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def test_shard_range_partition():
    from spiral_amd import dist as sdist

    for world in (1, 2, 4, 8):
        rs = [sdist.shard_range(r, world, 256) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == 256 and all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        sdist.shard_range(0, 3, 256)
    with pytest.raises(ValueError):
        sdist.shard_range(0, 32, 256)  # beyond the carry-safe bound of the packed sum

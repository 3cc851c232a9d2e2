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


def _worker_sharded_fold(rank, world, port, q, stages=1):
    """reduce-scatter + local fold + all-gather + root fold (bench.py's N > 1 path) with the oracle as the kernels"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import ctypes as C

    import torch
    import torch.distributed as dist

    from oracle import pyoracle as O
    from spiral_amd import dist as sdist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    po = O.make_params(2, 2, t_gsw=4)
    s = O.shape_of(po)
    cl = O.Client(po, seed=5)  # same keys and query on both ranks
    wl, wr, w, v = cl.pub_params()
    qy = cl.query(9)
    db = O.gen_db(po, 3)
    cv = O.stage_expand(po, qy, wl, wr)
    cts, gsw = O.stage_convert(po, cv, w, v)
    G = sdist.fold_ranks(world, s.num_per)
    j0, j1 = sdist.shard_range(rank, world, s.dim0)
    dbv = db.reshape(O.N, s.num_per, 2, s.dim0, 2)[:, :, :, j0:j1, :]
    part = O.multiply_query_by_database(O.reorient_ciphertexts(cts[j0:j1]), np.ascontiguousarray(dbv).reshape(-1), j1 - j0, s.num_per)
    packed = (part[..., 0, :] | (part[..., 1, :] << np.uint64(32))).astype(np.uint64)  # [ii][3][2][N]
    perm = np.array([sdist.acc_position(ii, G, s.num_per, stages) for ii in range(s.num_per)])  # Server.set_fold_ranks / set_sweep_stages layout
    assert sorted(perm) == list(range(s.num_per))
    if stages == 1:
        assert all(perm[ii] == (ii % G) * (s.num_per // G) + ii // G for ii in range(s.num_per))
    grouped = np.zeros_like(packed)
    grouped[perm] = packed
    acc = torch.from_numpy(grouped.view(np.int64).reshape(-1).copy())
    chunk = torch.zeros(acc.numel() // G, dtype=torch.int64)
    if stages == 1:
        sdist.reduce_scatter_accumulators(chunk, acc)
    else:  # one reduce-scatter per stage, each over a contiguous 1/stages of the buffer
        sdist.reduce_scatter_stages(chunk, acc, stages)
    L = s.num_per // G
    tot = chunk.numpy().view(np.uint64).reshape(L, 3, 2, O.N)
    ntt = np.stack([(tot & np.uint64(0xFFFFFFFF)) % np.uint64(O.P), (tot >> np.uint64(32)) % np.uint64(O.B)], axis=-2)
    raw = O.from_ntt(ntt)  # fold_local: lift, then nu2 - log2(G) rounds

    def fold_rounds(raw_cts, d0, rounds):
        x = np.ascontiguousarray(raw_cts).copy()
        npr = x.shape[0]
        m2 = s.m2
        for d in range(d0, d0 + rounds):
            npr //= 2
            qraw = O.from_ntt(gsw[d])
            g2 = O.build_gadget(3, m2)
            neg = ((g2.astype(object) - qraw.astype(object)) % O.Q).astype(np.uint64)
            q_re, qn_re = np.zeros(O.N * 3 * m2, dtype=np.uint64), np.zeros(O.N * 3 * m2, dtype=np.uint64)
            O.lib().orc_reorient_Q(O._p(q_re), O._p(np.ascontiguousarray(gsw[d])), C.c_uint32(m2))
            O.lib().orc_reorient_Q(O._p(qn_re), O._p(O.to_ntt(neg)), C.c_uint32(m2))
            O.lib().orc_fold_one_further_dimension(O._p(x), C.c_size_t(npr), O._p(q_re), O._p(qn_re), C.c_uint32(po.t_gsw))
        return x[0]

    k = int(np.log2(G))
    ct = torch.from_numpy(fold_rounds(raw, 0, po.nu2 - k).view(np.int64).reshape(-1).copy())
    gathered = torch.zeros(G * ct.numel(), dtype=torch.int64)
    sdist.all_gather_cts(gathered, ct)
    if rank == 0:
        fin = fold_rounds(gathered.numpy().view(np.uint64).reshape(G, 3, 2, O.N), po.nu2 - k, k)
        q.put(bool((fin == O.answer(po, qy, wl, wr, w, v, db)).all()))
    dist.destroy_process_group()


@pytest.mark.parametrize("stages", [1, 2])
def test_two_rank_distributed_fold_equals_single_device(stages):
    """stages = 2: the pipelined layout -- the accumulators [stage][rank][ct], one reduce-scatter per stage"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded_fold, args=(r, 2, port, q, stages)) for r in range(2)]
    for p in procs:
        p.start()
    ok = q.get(timeout=300)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert ok


def _worker_bad_tensors(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from spiral_amd import dist as sdist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    acc = torch.zeros(64, dtype=torch.int64)
    caught = []
    for name, fn in (
        ("dtype", lambda: sdist.reduce_scatter_accumulators(torch.zeros(32, dtype=torch.int64), acc.to(torch.int32))),
        ("numel", lambda: sdist.reduce_scatter_accumulators(torch.zeros(16, dtype=torch.int64), acc)),
        ("stride", lambda: sdist.reduce_accumulators(torch.zeros(128, dtype=torch.int64)[::2])),
        ("gather", lambda: sdist.all_gather_cts(torch.zeros(100, dtype=torch.int64), torch.zeros(64, dtype=torch.int64))),
        ("stages", lambda: sdist.reduce_scatter_stages(torch.zeros(32, dtype=torch.int64), acc, 3)),
    ):
        try:
            fn()
        except ValueError:
            caught.append(name)
    # and the well-formed call still goes through after the refusals (nothing was half-issued)
    chunk = torch.zeros(32, dtype=torch.int64)
    sdist.reduce_scatter_accumulators(chunk, acc + rank + 1)
    q.put((rank, caught, int(chunk[0])))
    dist.destroy_process_group()


def test_collectives_refuse_malformed_tensors():
    """spiral_amd/dist.py checks dtype, contiguity and word count of every tensor it hands to a collective BEFORE issuing it (a wrong view would not
    fail inside RCCL: it would reduce the wrong bytes, or hang every rank on a size mismatch); a refused call leaves the group usable"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_bad_tensors, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
    for rank, caught, first in got:
        assert caught == ["dtype", "numel", "stride", "gather", "stages"], (rank, caught)
        assert first == 3  # (0 + 1) + (1 + 1)


def test_shard_range_partition():
    from spiral_amd import dist as sdist

    for world in (1, 2, 4, 8):
        rs = [sdist.shard_range(r, world, 256) for r in range(world)]
        assert rs[0][0] == 0 and rs[-1][1] == 256 and all(rs[i][1] == rs[i + 1][0] for i in range(world - 1))
    with pytest.raises(ValueError):
        sdist.shard_range(0, 3, 256)
    with pytest.raises(ValueError):
        sdist.shard_range(0, 32, 256)  # beyond the carry-safe bound of the packed sum


def _expand_shard_with_oracle(O, po, s, cv, wl, wr, rank, G):
    """expandImproved (src/spiral.cpp:1664-1743) restricted to what rank `rank` of G needs -- the even subtree above its own
    first-dimension block and every G-th odd (GSW bit) ciphertext -- with the oracle's primitives as the kernels: the
    schedule of spiral_amd/csrc/host_common.h run_expand (ExpandShard), restated"""
    k, j_log = int(np.log2(G)), po.nu1 - int(np.log2(G))
    wl = wl.reshape(s.n_left, 2, po.t_exp, 2, O.N)
    wr = wr.reshape(s.n_right, 2, po.t_exp_right, 2, O.N)
    max_bits = s.ell * po.nu2
    for r in range(s.g):
        num_in, t = 1 << r, (O.N >> r) + 1
        raw = np.zeros((1, O.N), dtype=np.uint64)
        raw[0, O.N - num_in] = 1
        neg1 = O.to_ntt(O.invert(raw))[0]
        odd_total = 0 if r > s.stopround else (min(num_in, max_bits + 1) if r == s.stopround else num_in)
        if r > j_log:
            off = (rank & ((1 << (r - j_log)) - 1)) << j_log
            even = [2 * (a + off) for a in range(1 << j_log)]
        else:
            even = [2 * a for a in range(num_in)]
        odd = [2 * a + 1 for a in range(odd_total) if r < k or a % G == rank]
        active = even + odd
        for i in active:  # the ciphertexts this round creates (:1709), wherever this rank still holds the parent
            if i >= num_in:
                cv[i] = O.mul_by_const(neg1, cv[i - num_in])
        for i in active:
            gdim, W = (po.t_exp_right, wr[r]) if i & 1 else (po.t_exp, wl[r])
            ca = O.automorph(O.from_ntt(cv[i]), t)
            g_ntt = O.to_ntt(O.gadget_invert(ca[:1].reshape(1, 1, O.N), gdim, 1).reshape(gdim, O.N), reduce=False).reshape(gdim, 1, 2, O.N)
            upd = O.multiply(np.ascontiguousarray(W), g_ntt)
            upd[1] = O.add(upd[1], O.to_ntt(ca[1:2]))
            cv[i] = O.add(cv[i], upd[:, 0])
    return cv


def _worker_sharded_expansion(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from oracle import pyoracle as O
    from spiral_amd import dist as sdist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    kw = dict(t_gsw=4)
    po = O.make_params(3, 2, **kw)
    s = O.shape_of(po)
    assert s.stopround > 0 and sdist.expand_shard_ok(s, po, world)
    cl = O.Client(po, seed=5)
    wl, wr, w, v = cl.pub_params()
    qy = cl.query(13)
    cv = np.zeros((1 << s.g, 2, 2, O.N), dtype=np.uint64)
    cv[0] = qy.reshape(2, 2, O.N)
    cv = _expand_shard_with_oracle(O, po, s, cv, wl, wr, rank, world)
    n_bits = s.ell * po.nu2
    n_max = (n_bits + world - 1) // world
    mine = np.zeros((n_max, 2, 2, O.N), dtype=np.uint64)  # gsw_bits_pack: bit i = a * world + rank
    for a in range(n_max):
        if a * world + rank < n_bits:
            mine[a] = cv[2 * (a * world + rank) + 1]
    gathered = torch.zeros(world * mine.size, dtype=torch.int64)
    sdist.all_gather_gsw_bits(gathered, torch.from_numpy(mine.view(np.int64).reshape(-1).copy()))
    blocks = gathered.numpy().view(np.uint64).reshape(world, n_max, 2, 2, O.N)
    for r in range(world):  # gsw_bits_unpack
        for a in range(n_max):
            if a * world + r < n_bits:
                cv[2 * (a * world + r) + 1] = blocks[r, a]
    want = O.stage_expand(po, qy, wl, wr)  # [first-dimension cts | GSW bits], reordered
    j0, j1 = sdist.shard_range(rank, world, s.dim0)
    ok = all((cv[2 * j] == want[j]).all() for j in range(j0, j1)) and all((cv[2 * i + 1] == want[s.dim0 + i]).all() for i in range(n_bits))
    q.put((rank, bool(ok)))
    dist.destroy_process_group()


def test_two_rank_sharded_expansion_equals_full_expansion():
    """each rank expands its own subtree + every second GSW bit, one all-gather of the bits: every rank ends up with exactly
    the ciphertexts of the full expansion it needs (its own first-dimension block, all GSW bits)"""
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_sharded_expansion, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {0: True, 1: True}


def _worker_factor_shard(rank, world, port, q):
    """factor-sharded items (bench.py --workload stream --gpus N) with the oracle as the kernels: an item = factor 3 database instances, rank r answers
    the one query against instances r, r + world, ...; ONE all-gather of the responses; every rank can then decode the whole item"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist

    from oracle import pyoracle as O
    from spiral_amd import dist as sdist

    dist.init_process_group("gloo", rank=rank, world_size=world)
    kw = dict(t_gsw=5, t_exp=2, qprime_bits=19, direct_upload=1)  # the SpiralStream form: no expansion
    po = O.make_params(3, 2, **kw)
    s = O.shape_of(po)
    factor, idx = 3, 21
    cl = O.Client(po, seed=8)  # the same client (keys, query) on both ranks
    pp = cl.pub_params()
    qy = cl.query(idx)
    mine_k = sdist.instances_of_rank(rank, world, factor)
    assert mine_k == ([0, 2] if rank == 0 else [1])
    slots = (factor + world - 1) // world
    mine = torch.zeros(slots * 6 * O.N, dtype=torch.int64)
    for sl, k in enumerate(mine_k):
        fin = O.answer(po, qy, *pp, O.gen_db(po, 100 + k))  # instance k = the database seeded 100 + k
        mine[sl * 6 * O.N:(sl + 1) * 6 * O.N] = torch.from_numpy(O.stage_rescale(po, fin).view(np.int64).reshape(-1).copy())
    gathered = torch.zeros(world * mine.numel(), dtype=torch.int64)
    sdist.all_gather_instance_responses(gathered, mine)
    ok = True
    for k in range(factor):
        resp = sdist.instance_response(gathered, k, world, slots).numpy().view(np.uint64).reshape(3, 2, O.N)
        ok = ok and bool((cl.decode(resp) == O.db_item(po, 100 + k, idx)).all())
    q.put((rank, ok))
    dist.destroy_process_group()


def test_two_rank_factor_shard_gathers_every_instance():
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_factor_shard, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = dict(q.get(timeout=300) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert got == {0: True, 1: True}


def test_bench_prints_its_line_when_the_launcher_terminates_it(tmp_path):
    """bench.py's last net (Progress.catch_sigterm): a launcher that loses one rank SIGTERMs the others while their main threads sit in a collective, where no
    Python-level signal handler can run; the signal reaches the watchdog thread through a wake-up pipe and rank 0 prints the line-so-far before it exits"""
    import json
    import signal
    import subprocess
    import time

    code = (
        "import sys, ctypes\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "prog = bench.Progress(0, 100)\n"
        "prog.catch_sigterm()\n"
        "prog.update({'metric': 'm', 'value': 1.5, 'schedules': {'ms_per_query': {'in-order': 1.5}}})\n"
        "prog.arm('config2/comm-overlap', 1.0)\n"
        "print('ready', file=sys.stderr, flush=True)\n"
        "while True:\n"
        "    ctypes.CDLL(None).sleep(60)  # a blocking C call on the main thread, restarted when interrupted: what a collective's wait does\n"
    )
    p = subprocess.Popen([sys.executable, "-c", code], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=dict(os.environ, SPIRAL_BENCH_PARTIAL=str(tmp_path / "partial.json")))
    while "ready" not in p.stderr.readline():
        assert p.poll() is None
    time.sleep(0.3)
    p.send_signal(signal.SIGTERM)
    out, _ = p.communicate(timeout=30)
    assert p.returncode == 143
    line = json.loads(out.strip().splitlines()[-1])
    assert line["partial"] is True and line["terminated_by"] == "SIGTERM" and line["terminated_in"] == "config2/comm-overlap" and line["value"] == 1.5


def test_bench_failed_rank_prints_the_line_so_far_with_the_error():
    """an exception on rank 0 itself (bench.main's outer net): the line-so-far goes to stdout with `error` and `failed_in`, exit code 1"""
    import json
    import subprocess

    code = (
        "import sys\n"
        f"sys.path.insert(0, {ROOT!r})\n"
        "import bench\n"
        "def boom(args, prog, np):\n"
        "    prog.update({'metric': 'm', 'value': 2.5})\n"
        "    prog.arm('config2/pipelined', 1.0)\n"
        "    raise RuntimeError('device lost')\n"
        "bench.run = boom\n"
        "bench.main(['--no-cpu-baseline'])\n"
    )
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=120, env=dict(os.environ, SPIRAL_BENCH_PARTIAL=os.devnull))
    assert r.returncode == 1
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["partial"] is True and "device lost" in line["error"] and line["failed_in"] == "config2/pipelined" and line["value"] == 2.5

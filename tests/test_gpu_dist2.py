"""The N > 1 answer path as two real processes on one MI355X (both ranks on cuda:0, gloo collectives on device tensors):
j-shards, run_pre_sweep / fold_local / fold_root replayed as hipGraphs, reduce-scatter and all-gather between them, exactly the
call sequence of bench.py's step().  Checks the multi-process ordering between the library's stream work and
torch.distributed that the single-process emulation (test_gpu_parity.py::test_distributed_fold_emulated_on_one_gpu) cannot."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, root_fold, shard_expand, overlap, q, nu=(4, 4), stages=0):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch

    torch.cuda.is_available()  # torch initialises HIP first
    import torch.distributed as dist

    import spiral_amd as sa
    from oracle import pyoracle as O
    from spiral_amd import dist as sdist
    from spiral_amd import server as SV

    dist.init_process_group("gloo", rank=rank, world_size=world)
    dev = torch.device("cuda", 0)
    kw = dict(t_gsw=4)
    po, pg = O.make_params(*nu, **kw), sa.make_params(*nu, **kw)
    s = O.shape_of(po)
    cl = O.Client(po, seed=77)  # same keys and queries on every rank
    wl, wr, w, v = cl.pub_params()
    j0, j1 = sdist.shard_range(rank, world, s.dim0)
    srv = sa.Server(pg, 0, j0, j1)
    stream = torch.cuda.Stream(device=dev)
    srv.set_stream(stream.cuda_stream)
    srv.gen_db(5)
    srv.set_pub_params(wl, wr, w, v)
    words = s.num_per * 6 * sa.N
    acc = torch.zeros(words, dtype=torch.int64, device=dev)
    srv.set_acc(acc.data_ptr())
    G = 1 if root_fold else world
    srv.set_fold_ranks(G)
    chunk = torch.zeros(words // G, dtype=torch.int64, device=dev)
    ct = torch.zeros(6 * sa.N, dtype=torch.int64, device=dev)
    gathered = torch.zeros(G * 6 * sa.N, dtype=torch.int64, device=dev)
    if shard_expand:  # each rank expands its own subtree and every world-th GSW bit; the bits are all-gathered
        srv.set_expand_shard(rank, world)
        bits = torch.zeros(srv.gsw_bits_words(), dtype=torch.int64, device=dev)
        bits_all = torch.zeros(world * bits.numel(), dtype=torch.int64, device=dev)
    if stages:
        assert stages <= srv.max_sweep_stages()
        srv.set_sweep_stages(stages)  # accumulators laid out [stage][rank][ct]
    srv.use_graphs(True)
    ok = True
    db = O.gen_db(po, 5) if rank == 0 else None
    total = 1 << sum(nu)
    with torch.cuda.stream(stream):
        for idx in (9, total - 56, total - 1, 9):
            qy = cl.query(idx)
            srv.set_query(qy)
            if shard_expand and overlap and stages:  # bench.py's pipelined order: stage k's reduce-scatter under stage k + 1's sweep
                srv.run_expand_pack(bits.data_ptr())
                w_bits = sdist.all_gather_gsw_bits(bits_all, bits, async_op=True)
                srv.run_scal2mat()
                al, cl_ = acc.numel() // stages, chunk.numel() // stages
                works = []
                for k in range(stages):
                    srv.first_dim_stage(k)
                    works.append(sdist.reduce_scatter_accumulators(chunk[k * cl_:(k + 1) * cl_], acc[k * al:(k + 1) * al], async_op=True))
                w_bits.wait()
                srv.run_unpack_gsw(bits_all.data_ptr())
                for wk in works:
                    if wk is not None:
                        wk.wait()
            elif shard_expand and overlap:  # bench.py's overlapped order: the all-gather under ScalToMat + sweep
                srv.run_expand_pack(bits.data_ptr())
                w_bits = sdist.all_gather_gsw_bits(bits_all, bits, async_op=True)
                srv.run_scal2mat_sweep()
                w_acc = sdist.reduce_scatter_accumulators(chunk, acc, async_op=True)
                w_bits.wait()
                srv.run_unpack_gsw(bits_all.data_ptr())
                if w_acc is not None:
                    w_acc.wait()
            elif shard_expand:
                srv.run_expand_pack(bits.data_ptr())
                sdist.all_gather_gsw_bits(bits_all, bits)
                srv.run_unpack_convert_sweep(bits_all.data_ptr())
            else:
                srv.run_pre_sweep()
            if root_fold:
                sdist.reduce_accumulators(acc, dst=0)
                if rank == 0:
                    srv.run_post(reduce_first=True)
            else:
                if not (shard_expand and overlap):
                    sdist.reduce_scatter_accumulators(chunk, acc)
                srv.fold_local(chunk.data_ptr(), ct.data_ptr())
                sdist.all_gather_cts(gathered, ct)
                if rank == 0:
                    srv.fold_root(gathered.data_ptr())
            srv.sync()
            if rank == 0:
                want = O.answer(po, qy, wl, wr, w, v, db)
                got = srv.read(SV.BUF_FINAL)
                ok = ok and bool((got == want).all()) and bool((cl.decode(srv.read(SV.BUF_RESPONSE)) == O.db_item(po, 5, idx)).all())
            dist.barrier()
    srv.close()
    if rank == 0:
        q.put(ok)
    dist.destroy_process_group()


def _run_ranks(world, *args, **kw):
    import torch.multiprocessing as mp

    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port) + args + (q,), kwargs=kw) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=600)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert q.get(timeout=5) is True


@pytest.mark.parametrize("root_fold,shard_expand,overlap", [(False, False, False), (True, False, False), (False, True, False), (True, True, False), (False, True, True)])
def test_two_processes_one_gpu(root_fold, shard_expand, overlap):
    _run_ranks(2, root_fold, shard_expand, overlap)


def test_two_processes_pipelined_sweep():
    """the pipelined schedule with two real processes: num_per = 64 ciphertexts swept in 2 stages, each stage's half of the accumulator
    buffer reduce-scattered on its own (async) while the next stage runs; answer == the oracle's"""
    _run_ranks(2, False, True, True, nu=(5, 6), stages=2)


@pytest.mark.parametrize("world,nu,stages", [(4, (4, 4), 0), (8, (4, 4), 0), (4, (5, 6), 2), (8, (6, 6), 2)])
def test_four_and_eight_processes_one_gpu(world, nu, stages):
    """the answer path with 4 and 8 REAL processes (every rank on cuda:0, gloo collectives): sharded expansion (each rank its own subtree
    and every world-th GSW bit), j-shards of dim0 / world, reduce-scatter by ciphertext, local folds, all-gather, root folds -- the
    comm-overlap order, and the pipelined order (sweep stages) -- bit-exact against the oracle on rank 0"""
    _run_ranks(world, False, True, True, nu=nu, stages=stages)


def test_bench_two_rank_flow_on_one_gpu():
    """bench.py's own N = 2 step loop (sampled stage events, run_pre_sweep, the collectives, fold_local / fold_root), launched the
    way the driver launches it, with the two ranks sharing the one device and gloo standing in for RCCL: one JSON line from rank 0"""
    import json
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--backend", "gloo", "--shared-device", "--prewarm", "2", "--no-config3"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["roofline"]["bound"] == "hbm" and out["roofline"]["achieved"] > 0
    assert "reduce-scatter" in out["config"]["parallelism"] and "sharded expansion" in out["config"]["parallelism"]


def test_bench_self_launch_without_launcher():
    """plain `python bench.py --gpus 2` with no WORLD_SIZE in the environment: bench.py starts the two ranks itself (before it
    touches the GPU), relays rank 0's JSON line and exits with the children's code.  The line is self-describing: what the
    communicator saw, both collective schedules timed over the same steps, per-collective times, the configs[2] leg."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--backend", "gloo", "--shared-device", "--prewarm", "2", "--config3-steps", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["value"] > 0
    assert out["rccl"]["world_size"] == 2 and [x["rank"] for x in out["rccl"]["ranks_seen"]] == [0, 1] and out["rccl"]["backend"] == "gloo"
    assert set(out["schedules"]["ms_per_query"]) == {"in-order", "comm-overlap", "pipelined"} and out["schedules"]["chosen"] == "comm-overlap"
    assert out["value"] == out["schedules"]["ms_per_query"]["comm-overlap"] and out["schedules"]["sweep_stages"] == 4
    assert out["schedules"]["fastest"] in out["schedules"]["ms_per_query"]
    assert set(out["collectives_us"]) == {"all_gather_gsw_bits", "reduce_scatter_accumulators", "all_gather_folded_cts"}
    c3 = out["also"]["config3"]
    assert c3["n_gpus"] == 2 and c3["value"] > 0 and c3["roofline"]["achieved"] > 0 and "2^24" in c3["workload"]
    # the labelled throughput block: every rank batches whole queries on its own full copy of the database (never `value`)
    assert out["replicas"]["n_replicas"] == 2 and out["replicas"]["batch"] == 8 and out["replicas"]["queries_per_s"] > 0
    assert len(out["answer_sha256"]) == 64 and "partial" not in out


def test_bench_eight_ranks_on_one_gpu():
    """`python bench.py --gpus 8` on the launcher-less path with eight ranks sharing the one device (gloo): the world the first 8-GPU run
    will have -- eight ranks seen by the communicator, all three schedules timed, rank 0's line"""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--shared-device", "--prewarm", "2", "--nu1", "7", "--nu2", "6", "--no-config3"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["rccl"]["world_size"] == 8 and sorted(x["rank"] for x in out["rccl"]["ranks_seen"]) == list(range(8))
    assert set(out["schedules"]["ms_per_query"]) == {"in-order", "comm-overlap", "pipelined"} and out["schedules"]["sweep_stages"] == 2
    assert set(out["collectives_us"]) == {"all_gather_gsw_bits", "reduce_scatter_accumulators", "all_gather_folded_cts"}
    assert "j-shard x8" in out["config"]["parallelism"] and out["value"] > 0


def test_bench_watchdog_keeps_the_finished_schedules(tmp_path):
    """a hang in the last (and least hardware-tested) schedule must not lose the run: with SPIRAL_BENCH_INJECT_HANG=pipelined both ranks stop inside
    that schedule; the watchdog prints the line-so-far -- in-order and comm-overlap timed, what the communicator saw, the collective times -- on
    stdout, marked partial, and the job exits non-zero within the watchdog's time; the same line is in the partial file"""
    import json
    import subprocess
    import time

    partial = tmp_path / "partial.json"
    env = dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")},
               SPIRAL_BENCH_INJECT_HANG="pipelined", SPIRAL_BENCH_WATCHDOG_S="20", SPIRAL_BENCH_PARTIAL=str(partial))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--backend", "gloo", "--shared-device", "--prewarm", "2", "--nu1", "7", "--nu2", "6", "--no-config3"]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert r.returncode != 0, "the injected hang must end in a non-zero exit"
    assert time.time() - t0 < 300
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["partial"] is True and out["hung_in"] == "config2/pipelined" and out["n_gpus"] == 2
    assert set(out["schedules"]["ms_per_query"]) == {"in-order", "comm-overlap"} and out["schedules"]["requested"] == ["in-order", "comm-overlap", "pipelined"]
    assert out["value"] == out["schedules"]["ms_per_query"]["comm-overlap"] > 0
    assert out["rccl"]["world_size"] == 2 and set(out["collectives_us"]) == {"all_gather_gsw_bits", "reduce_scatter_accumulators", "all_gather_folded_cts"}
    assert out["roofline"]["achieved"] > 0 and len(out["answer_sha256"]) == 64
    assert out["schedules"]["answer_sha256"]["in-order"] == out["schedules"]["answer_sha256"]["comm-overlap"]
    on_disk = json.loads(partial.read_text())
    assert on_disk["partial"] is True and on_disk["schedules"]["ms_per_query"] == out["schedules"]["ms_per_query"]
    assert "bench.py partial: " in r.stderr


@pytest.mark.parametrize("extra", [[], ["--root-fold"]])
def test_bench_rccl_world_size_one(extra):
    """RCCL itself on hardware: bench.py with torch.distributed initialised on the nccl (= RCCL) backend and a world of one
    rank, so that init_process_group(device_id=...), reduce_scatter_tensor / all_gather_into_tensor / reduce on the int64
    accumulator tensors, the barrier and the max-over-ranks all-reduce all execute through RCCL, in the step loop the
    N-GPU runs use (the collectives are identities at world size 1; what is tested is that they run on this stack)"""
    import json
    import subprocess

    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "4", "--warmup", "1", "--force-dist", "--prewarm", "2", "--backend", "nccl", "--no-cpu-baseline", "--no-config3"] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["value"] > 0 and out["roofline"]["achieved"] > 0
    assert ("1 reduce" in out["config"]["parallelism"]) == bool(extra)
    if not extra:
        # all three schedules ran through RCCL, the pipelined one with its K asynchronous reduce-scatters (async_op=True) interleaved with the
        # sweep stages on the library's stream -- and every schedule left the same answer
        assert set(out["schedules"]["ms_per_query"]) == {"in-order", "comm-overlap", "pipelined"} and out["schedules"]["sweep_stages"] == 4
        assert len(set(out["schedules"]["answer_sha256"].values())) == 1 and out["rccl"]["backend"] == "nccl"
    assert "partial" not in out and len(out["answer_sha256"]) == 64


def test_bench_pack_trial_shards():
    """bench.py --workload pack on two ranks sharing the one device (gloo standing in for RCCL): each rank sweeps and folds 8 of the
    16 trials, the folded ciphertexts are all-gathered, rank 0 packs; and the same flow through RCCL with a world of one rank"""
    import json
    import subprocess

    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1", "--master-port",
           str(_free_port()), os.path.join(ROOT, "bench.py"), "--workload", "pack", "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--shared-device", "--prewarm", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["value"] > 0 and "trials x2 (8 per rank)" in out["config"]["parallelism"] and out["roofline"]["achieved"] > 0
    assert out["rccl"]["world_size"] == 2 and [x["rank"] for x in out["rccl"]["ranks_seen"]] == [0, 1] and out["collectives_us"]["all_gather_folded_trials"] > 0
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "pack", "--gpus", "1", "--steps", "3", "--warmup", "1", "--force-dist", "--prewarm", "2", "--backend", "nccl"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and "trials x1 (16 per rank)" in out["config"]["parallelism"] and out["roofline"]["frac"] > 0.5


def test_bench_watchdog_names_the_rank_that_never_joined():
    """the process-group set-up has its own short watchdog: with SPIRAL_BENCH_INJECT_HANG=pg-setup:1 rank 1 never calls init_process_group; rank 0's
    watchdog fires after --pg-watchdog seconds with a line that says which rank never checked in, and the job exits non-zero -- a fresh exit, nothing re-exec'd"""
    import json
    import subprocess
    import time

    env = dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")},
               SPIRAL_BENCH_INJECT_HANG="pg-setup:1", SPIRAL_BENCH_PG_WATCHDOG_S="20")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--backend", "gloo", "--shared-device", "--prewarm", "2", "--nu1", "7", "--nu2", "6", "--no-config3"]
    t0 = time.time()
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert r.returncode != 0 and time.time() - t0 < 240
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["partial"] is True and out["hung_in"] == "process group set-up" and out["watchdog_s"] == 20
    assert out["ranks_never_joined"] == [1] and out["ranks_checked_in"] == [0]
    assert "made no progress in 'process group set-up'" in r.stderr


def test_bench_collective_smoke_and_headline_swap():
    """two ranks on the one device (gloo): the collective smoke test ran and is in `rccl`; --headline config3 makes the 2^24 x 256 B geometry the `value`
    (here shrunk with --nu1/--nu2, which disables the secondary leg) -- and `config` carries the answer hash and the no-pre-warm value for the driver's record"""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--backend", "gloo", "--shared-device", "--prewarm", "2", "--headline", "config3",
           "--nu1", "7", "--nu2", "6"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["rccl"]["collective_smoke"]["ok"] is True and out["rccl"]["world_size"] == 2
    assert "configs[2]" in out["config"]["workload"] and "t_GSW=10" in out["config"]["workload"]
    assert out["config"]["answer_sha256"] == out["answer_sha256"] and len(out["answer_sha256"]) == 64
    assert out["config"]["value_no_prewarm"] == out["value_no_prewarm"]["value"] > 0


def test_bench_stream_item_factor_sharded_over_two_and_eight_ranks():
    """bench.py --workload stream: a 100 KB item = 7 database instances; two ranks (one device, gloo) hold instances {0, 2, 4, 6} and {1, 3, 5}, every rank
    converts the one query, ONE all-gather of the seven responses, no reduce.  At a geometry small enough for all seven instances to be resident on the one
    device; the same answer hash as the one-rank run (which sweeps the seven instances one after the other)."""
    import json
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "stream", "--steps", "3", "--warmup", "1", "--nu1", "6", "--nu2", "6"]
    r2 = subprocess.run(base + ["--gpus", "2", "--backend", "gloo", "--shared-device"], capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r2.returncode == 0, r2.stdout[-2000:] + r2.stderr[-2000:]
    o2 = json.loads([ln for ln in r2.stdout.splitlines() if ln.startswith("{")][-1])
    r1 = subprocess.run(base + ["--gpus", "1"], capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r1.returncode == 0, r1.stdout[-2000:] + r1.stderr[-2000:]
    o1 = json.loads([ln for ln in r1.stdout.splitlines() if ln.startswith("{")][-1])
    for o, world, mine in ((o2, 2, [0, 2, 4, 6]), (o1, 1, list(range(7)))):
        it = o["item"]
        assert o["n_gpus"] == world and it["factor"] == 7 and it["instances_of_rank0"] == mine and it["instances_resident"] == len(mine) == it["instances_swept_per_query_rank0"]
        assert it["db_device_bytes_rank0"] == len(mine) * it["image_bytes_per_instance"] and o["value"] > 0 and o["roofline"]["achieved"] > 0
    assert o1["answer_sha256"] == o2["answer_sha256"] == o2["config"]["answer_sha256"] and len(o1["answer_sha256"]) == 64
    assert "no reduce" in o2["config"]["parallelism"] and o2["rccl"]["collective_smoke"]["ok"] is True
    # N = 8, what the driver's scaling run does: one instance per rank and rank 7 holds NONE (factor 7) -- it only takes part in the all-gather
    r8 = subprocess.run(base + ["--gpus", "8", "--backend", "gloo", "--shared-device"], capture_output=True, text=True, cwd=ROOT, timeout=900, env=env)
    assert r8.returncode == 0, r8.stdout[-2000:] + r8.stderr[-2000:]
    o8 = json.loads([ln for ln in r8.stdout.splitlines() if ln.startswith("{")][-1])
    assert o8["n_gpus"] == 8 and o8["item"]["instances_of_rank0"] == [0] and o8["answer_sha256"] == o1["answer_sha256"] and "partial" not in o8


def _bench_env(**extra):
    return dict({k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}, **extra)


def test_bench_failed_extra_leg_does_not_cost_the_line():
    """a leg beside the headline that raises (an allocation failure in the whole-item leg, say) is recorded in the line -- `also.stream_item.error` -- and the
    line is still printed, complete otherwise, with exit code 0"""
    import json
    import subprocess

    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "4", "--warmup", "1", "--prewarm", "2", "--lanes", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=900, env=_bench_env(SPIRAL_BENCH_INJECT_FAIL="stream_item"))
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert "partial" not in out and out["value"] > 0 and len(out["answer_sha256"]) == 64
    assert "injected failure" in out["also"]["stream_item"]["error"]


def test_bench_rank_failure_still_prints_rank0s_line(tmp_path):
    """rank 1 raises in the second schedule and exits.  Rank 0 then either gets an error out of its collective (gloo: the peer closed the connection) -- bench.main's
    outer net prints the line-so-far with `error` -- or sits in the collective until the launcher's SIGTERM, which its watchdog thread sees through the wake-up pipe
    (`terminated_by`), or until the watchdog (`hung_in`): in every case the first schedule's timing reaches stdout"""
    import json
    import subprocess

    env = _bench_env(SPIRAL_BENCH_INJECT_FAIL="comm-overlap:1", SPIRAL_BENCH_PARTIAL=str(tmp_path / "partial.json"))
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--backend", "gloo", "--shared-device", "--prewarm", "2", "--nu1", "7", "--nu2", "6", "--no-config3"]
    r = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT, timeout=600, env=env)
    assert r.returncode != 0
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert lines, r.stderr[-3000:]
    out = json.loads(lines[-1])
    assert out["partial"] is True and out["n_gpus"] == 2 and (out.get("terminated_by") == "SIGTERM" or "hung_in" in out or "error" in out)
    assert set(out["schedules"]["ms_per_query"]) == {"in-order"} and out["value"] == out["schedules"]["ms_per_query"]["in-order"] > 0

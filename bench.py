#!/usr/bin/env python3
"""Headline benchmark: server ms/query and first-dimension sweep GB/s vs the HBM roofline on
BASELINE.json config 2 (Base Spiral, 2^20 x 256 B = nu1 8, nu2 7, the "(20, 256)" parameter set).

    python bench.py --gpus N --steps K --warmup W          (N > 1 under torch.distributed.run)

One step = the whole server-answer path for one query with database, public parameters and query
resident in HBM: coefficient expansion, ScalToMat + Regev->GSW conversion, the first-dimension sweep over
this rank's database shard, [one RCCL reduce of the per-shard accumulators], INTT + CRT lift, GSW
folding and the response modulus switch.  N ranks shard the SAME database by first-dimension index
(strong scaling); value = wall ms per query over the timed K steps, max over ranks.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E spec peak, /opt/skills/guides/MI355X_MICROARCH.md


def pmc_traffic(n_gpus, nu1, nu2):
    """HBM bytes per sweep launch from the committed rocprofv3 PMC passes (FETCH_SIZE / WRITE_SIZE, separate
    runs, corrected as MI355X_MICROARCH.md prescribes); only valid for the configuration it was taken on"""
    try:
        files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("sweep_pmc.json"))
        d = json.load(open(os.path.join(ROOT, "profiles", files[-1])))
        if n_gpus == 1 and (nu1, nu2) == (8, 7):
            return d["traffic_bytes_per_launch"], files[-1]
    except Exception:
        pass
    return None, None


def batch_roofline_from_profile(nu1, nu2):
    """per-kernel-class roofline of one eight-query batch (run_query_batch) from the committed rocprofv3 passes -- kernel-trace durations, FETCH_SIZE and
    WRITE_SIZE each in its own pass, corrected as MI355X_MICROARCH.md prescribes (tools/batch_roofline.sh, tools/batch_bytes.py).  Like roofline.traffic it
    is read from the file named in `source`, NOT measured by the run that prints it; valid for configs[1] on one GPU only."""
    try:
        files = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("batch_kernel_bytes_B8.json") and f.startswith("r"))
        if not files or (nu1, nu2) != (8, 7):
            return None
        d = json.load(open(os.path.join(ROOT, "profiles", files[-1])))
        return {"source": "profiles/" + files[-1], "lanes": d["lanes"], "batch_us": d["batch_us"], "peak_GBps": d["peak_GBps"], "per_kernel_class": d["per_kernel_class"],
                "note": "us from a rocprofv3 kernel trace of one batch; bytes = FETCH_SIZE x 1024 x 2 + WRITE_SIZE x 1024 from separate counter passes; not measured by this run"}
    except Exception:
        return None


def synth_residues(rng, np, shape):
    """uniform canonical NTT-form polynomials [..., 2, N] (synthetic query / public parameters)"""
    import spiral_amd as sa

    return np.stack([rng.integers(0, m, size=shape + (sa.N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)


def synth_inputs(np, sa, pg, shp):
    """the benchmark's synthetic public parameters and query: uniform canonical residues from default_rng(1), the same on every rank
    (tests/test_gpu_fullsize.py rebuilds exactly these to put the oracle's answer next to the timed graph's)"""
    rng = np.random.default_rng(1)
    pub = (synth_residues(rng, np, (max(shp.n_left, 1), 2, pg.t_exp)), synth_residues(rng, np, (max(shp.n_right, 1), 2, pg.t_exp_right)),
           synth_residues(rng, np, (3, 2 * pg.t_conv)), synth_residues(rng, np, (3, 2 * pg.t_conv)))
    query = synth_residues(rng, np, (shp.n_query_cts, 2))
    return pub, query


DB_SEED = 1234  # the explicit database every leg generates on the device (gen_db)


def answer_hash(np, final_ct, response):
    """sha256 over the folded ciphertext (3 x 2 x 2048 u64, little endian) followed by the switched response: what `answer_sha256` in the
    line is, and what the full-size GPU test computes from the oracle's answer on the same inputs"""
    import hashlib

    h = hashlib.sha256()
    h.update(np.ascontiguousarray(final_ct, dtype="<u8").tobytes())
    h.update(np.ascontiguousarray(response, dtype="<u8").tobytes())
    return h.hexdigest()


def inject_fail(tag, rank):
    """test hook: SPIRAL_BENCH_INJECT_FAIL=<tag>[:<rank>] raises at the point named <tag> (a schedule, "stream_item", "cpu_baseline"), on one rank or on all"""
    want = os.environ.get("SPIRAL_BENCH_INJECT_FAIL", "")
    if want and want in (tag, f"{tag}:{rank}"):
        raise RuntimeError(f"SPIRAL_BENCH_INJECT_FAIL={want}: injected failure on rank {rank}")


class Progress:
    """The line-so-far and a watchdog.  The first multi-GPU run may be the only one: after every schedule and every leg the line as it
    stands (same keys, "partial": true) goes to stderr and to bench_partial.json, and a watchdog thread per phase turns a hang (an RCCL
    collective that never completes, say) into that line on stdout and a non-zero exit instead of a lost run.  No process is re-exec'd."""

    def __init__(self, rank, seconds):
        import threading

        self.rank, self.seconds = rank, seconds
        self.line, self.where, self.deadline, self.limit = None, "set-up", None, seconds
        self.lock = threading.Lock()
        self.wrap = lambda line: line  # the secondary leg nests its line under the primary's `also`
        self.on_hang = None  # optional: () -> dict merged into the hung line (the process-group set-up says which ranks never joined)
        self.sig_r = None  # read end of the signal wake-up pipe (catch_sigterm)
        self.done = False  # the final line is out: a later SIGTERM must not print a second one
        t = threading.Thread(target=self._watch, daemon=True)
        t.start()

    def catch_sigterm(self):
        """A launcher that loses one rank (an exception, an out-of-memory kill) sends SIGTERM to the others: rank 0 then still owes its line.  The main thread
        may be inside a collective or a synchronize, where a Python-level handler never runs -- so the signal only writes to a wake-up pipe
        (signal.set_wakeup_fd, done by the C-level handler at once) and the watchdog thread, which polls the pipe, prints the line-so-far and exits."""
        import signal

        try:
            r, w = os.pipe()
            os.set_blocking(r, False)
            os.set_blocking(w, False)
            signal.signal(signal.SIGTERM, lambda signum, frame: None)  # (keeps the default action -- immediate death -- from happening)
            signal.set_wakeup_fd(w, warn_on_full_buffer=False)
            self.sig_r = r
        except (ValueError, OSError):  # not the main thread / no pipe: the run goes on without this net
            self.sig_r = None

    def give_up(self, code, **more):
        """rank 0: the line-so-far on stdout with `more` (why), then a fresh exit; other ranks just exit"""
        with self.lock:
            line, done = self.line, self.done
            self.done = True
        if self.rank == 0 and not done:
            out = dict(line or {"metric": "server ms/query + DB GB/s vs HBM roofline", "value": None}, partial=True, **more)
            sys.stdout.write(json.dumps(out) + "\n")
            sys.stdout.flush()
        os._exit(code)

    def arm(self, where, factor=1.0, seconds=None):
        """seconds: this phase's own limit (the process-group set-up's is short); otherwise `factor` x the --watchdog seconds"""
        limit = seconds if seconds is not None else self.seconds * factor
        with self.lock:
            self.where, self.limit, self.deadline = where, limit, (time.monotonic() + limit) if (self.seconds > 0 and limit > 0) else None

    def disarm(self):
        with self.lock:
            self.deadline = None

    def update(self, line):
        """record the line-so-far; rank 0 also writes it to stderr and to bench_partial.json (cwd), marked partial"""
        line = dict(self.wrap(line), partial=True)
        with self.lock:
            self.line = line
        if self.rank == 0:
            txt = json.dumps(line)
            print("bench.py partial: " + txt, file=sys.stderr, flush=True)
            try:
                with open(os.environ.get("SPIRAL_BENCH_PARTIAL", "bench_partial.json"), "w") as f:
                    f.write(txt + "\n")
            except OSError:
                pass

    def _watch(self):
        while True:
            time.sleep(0.25)
            if self.sig_r is not None:
                try:
                    got = os.read(self.sig_r, 64)
                except (BlockingIOError, OSError):
                    got = b""
                if 15 in got:  # SIGTERM
                    print(f"bench.py: rank {self.rank} received SIGTERM in '{self.where}' (another rank died and the launcher is tearing the job down?)", file=sys.stderr, flush=True)
                    self.give_up(143, terminated_by="SIGTERM", terminated_in=self.where)
            with self.lock:
                late = self.deadline is not None and time.monotonic() > self.deadline
                line, where, limit, hook = self.line, self.where, self.limit, self.on_hang
            if late:
                more = {}
                try:
                    more = hook() if hook else {}
                except Exception as e:  # the hook is diagnostics only
                    more = {"hang_diagnostics_failed": repr(e)}
                if self.rank == 0:
                    out = dict(line or {"metric": "server ms/query + DB GB/s vs HBM roofline", "value": None}, partial=True, hung_in=where,
                               watchdog_s=limit, **more)
                    sys.stdout.write(json.dumps(out) + "\n")
                    sys.stdout.flush()
                else:
                    time.sleep(10)  # rank 0 goes first: the launcher tears every rank down as soon as one exits
                print(f"bench.py: rank {self.rank} made no progress in '{where}' for {limit} s; giving up (exit 3, a fresh exit: nothing is re-exec'd) {more or ''}", file=sys.stderr, flush=True)
                os._exit(3)  # the main thread is stuck in a collective or a synchronize: no clean way out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_quota_cores():
    """CPUs this process may actually use: the cgroup quota (cpu.max) when there is one, else the affinity mask.  A GPU box reports
    256 logical CPUs but a 1-GPU slice of it is capped (16 CPUs on this pool): threads beyond the cap only time-slice."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(quota) // int(period)))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(params_kw, np, workload):
    """the oracle (CPU restatement of the reference's algorithm, kind "port") timed on this box's host cores on ONE full
    query of the same workload, twice: on 1 thread -- the reference is single-threaded (src/spiral.cpp:1231, no -fopenmp in
    CMakeLists.txt:11) -- and on all cores (OpenMP over NTT slots in the sweep and over polynomials / ciphertexts in the
    transforms).  Built here, on the machine it is timed on, with -O3 -march=native (oracle/Makefile
    `native`).  The database is arbitrary valid NTT-form words since only timing matters."""
    import shutil
    import subprocess
    import tempfile

    from oracle import pyoracle as O

    d = tempfile.mkdtemp(prefix="oracle_native_")
    for f in ("spiral_oracle.c", "spiral_oracle_pack.c", "spiral_oracle.h", "Makefile"):
        shutil.copy(os.path.join(ROOT, "oracle", f), d)
    native = True
    try:
        subprocess.check_call(["make", "-C", d, "-s", "native"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        O.LIB_PATH = os.path.join(d, "liboracle_native.so")
    except Exception:
        native = False  # no compiler on this box: the prebuilt x86-64-v3 single-threaded library
        O.build()
    po = O.make_params(**params_kw)
    s = O.shape_of(po)
    rng = np.random.default_rng(7)
    db = O.fill_db_random(99, O.db_words(po))
    mk = lambda shape: np.ascontiguousarray(np.stack([rng.integers(0, m, size=shape + (O.N,), dtype=np.uint64) for m in (O.P, O.B)], axis=-2))
    wl, wr = mk((max(s.n_left, 1), 2, po.t_exp)), mk((max(s.n_right, 1), 2, po.t_exp_right))
    w, v = mk((3, 2 * po.t_conv)), mk((3, 2 * po.t_conv))
    q = mk((s.n_query_cts, 2))

    sweep_bytes = O.db_words(po) * 8  # the NTT-form database the loop streams, 8 B per word as SURVEY.md 8d counts it

    def timed(threads, reps=1):
        """best of `reps` passes of the four stages (expansion, conversion, first dimension = reorient + sweep + lift, folding):
        (threads, total ms, first-dimension ms, sweep-loop ms)"""
        got, best = O.set_threads(threads), None
        for _ in range(reps):
            t0 = time.perf_counter()
            cv = O.stage_expand(po, q, wl, wr)
            cts, gsw = O.stage_convert(po, cv, w, v)
            t1 = time.perf_counter()
            re = O.reorient_ciphertexts(cts)
            t2 = time.perf_counter()
            acc = O.multiply_query_by_database(re, db, s.dim0, s.num_per)
            t3 = time.perf_counter()
            raw = O.from_ntt(acc)
            t4 = time.perf_counter()
            O.stage_rescale(po, O.stage_fold(po, raw, gsw))
            t5 = time.perf_counter()
            cur = ((t5 - t0) * 1e3, (t4 - t1) * 1e3, (t3 - t2) * 1e3)
            if best is None or cur[0] < best[0]:
                best = cur
        return (got,) + best

    # 1 thread: once with the scalar restatement of the transforms (reported beside), then -- `value` -- with the reference's USE_AVX2
    # form of them (forward butterflies four at a time for t >= 4 + vector closing corrections, src/core.cpp:292-349, 479-506;
    # tests/test_oracle.py proves the two equal), so that the baseline is the reference's algorithm AND its instruction mix
    _, ms_scalar, _, _ = timed(1)
    simd = O.set_ntt_simd(True)
    _, ms1, fd1, sw1 = timed(1)
    out = {"value": round(ms1, 1), "unit": "ms/query", "cores": 1, "kind": "port", "cpu_model": cpu_model(),
           "transform_isa": (O.ntt_isa() + " forward butterflies (t >= 4) and closing corrections, scalar inverse butterflies -- the reference's USE_AVX2 split") if simd else "scalar",
           "value_scalar_transforms": round(ms_scalar, 1),
           "first_dim_ms": round(fd1, 1), "sweep_loop_ms": round(sw1, 1), "sweep_gbps": round(sweep_bytes / sw1 / 1e6, 2), "sweep_isa": O.sweep_isa(),
           "build": "gcc -O3 -march=native -fopenmp on this box" if native else "prebuilt gcc -O3 -march=x86-64-v3",
           "sample": f"1 full query of {workload} ({O.db_words(po) * 8 / 2**30:.0f} GiB NTT-form database of arbitrary valid words), oracle/ restatement; "
                     "the first-dimension loop is the reference's vectorised form (src/spiral.cpp:640-886, _mm512_mul_epu32 / _mm256_mul_epu32, partial "
                     "reduction every 64 terms); sweep_gbps = NTT-form database bytes / sweep_loop_ms, comparable with roofline.achieved"}
    if native:
        # all cores: a thread ladder, best of two passes each (the second is warm), the best count reported
        ncpu, quota = os.cpu_count() or 1, cpu_quota_cores()
        O.set_threads(min(quota, 16))
        db = O.fill_db_random(99, O.db_words(po))  # first touch spread over the threads' NUMA nodes
        ladder, best = {}, None
        for th in sorted({t for t in (8, 16, 32, 64, 128, quota, 2 * quota) if t <= min(ncpu, 2 * quota)}):
            got, ms, fd, sw = timed(th, reps=2)
            ladder[str(got)] = round(ms, 1)
            if best is None or ms < best[1]:
                best = (got, ms, fd, sw)
            elif ms > 3 * best[1]:
                break
        out["all_cores"] = {"value": round(best[1], 1), "unit": "ms/query", "cores": best[0], "logical_cpus": ncpu, "usable_cpus": quota, "first_dim_ms": round(best[2], 1),
                            "sweep_loop_ms": round(best[3], 1), "sweep_gbps": round(sweep_bytes / best[3] / 1e6, 2), "ms_by_threads": ladder,
                            "note": "usable_cpus = the cgroup CPU quota of this box (cpu.max) or the affinity mask: the thread ladder stops at twice that"}
    shutil.rmtree(d, ignore_errors=True)
    return out


# BASELINE.json configs that run through the base server (bench.py --workload); parameter sets: all_parameter_choices.txt
WORKLOADS = {
    # configs[1] (= configs[0]'s geometry), "(20, 256)/spiral", all_parameter_choices.txt:67-81 -- the configuration the metric is quoted on
    "config2": dict(nu1=8, nu2=7, t_gsw=8, t_conv=4, t_exp=8, t_exp_right=56, qprime_bits=20, p_db=256,
                    label="configs[1]: Base Spiral 2^20 x 256B (nu1=8, nu2=7, p=256, t_GSW=8, t_conv=4, t_exp=8, t_exp_right=56, q'=2^20)"),
    # configs[2]: 2^24 x 256 B = 32 GiB in the reference's layout; no published set, SURVEY.md 8d's choice
    "config3": dict(nu1=9, nu2=10, t_gsw=10, t_conv=4, t_exp=8, t_exp_right=56, qprime_bits=22, p_db=256,
                    label="configs[2]: Base Spiral 2^24 x 256B (nu1=9, nu2=10, p=256, t_GSW=10, t_conv=4, t_exp=8, t_exp_right=56, q'=2^22)"),
    # configs[3]: "Streaming 20/spiralstream", all_parameter_choices.txt:1149-1163, one 64 GiB instance
    "stream": dict(nu1=11, nu2=9, t_gsw=4, t_conv=56, t_exp=2, t_exp_right=56, qprime_bits=27, p_db=32768, direct_upload=1,
                   label="configs[3]: SpiralStream (--direct-upload) 2^20 x 100KB (nu1=11, nu2=9, p=32768, t_GSW=4, t_conv=56, q'=27 bits)"),
}


def bench_pack(args):
    """--workload pack: BASELINE.json configs[4], SpiralPack 2^18 x 30 KB (all_parameter_choices.txt "(18, 30000)"/spiral-pack:
    nu1=10, nu2=8, n=4, p=256, q'=2^20, t_GSW=8, t_conv=4, t_exp=16) -- 16 trial databases of 4 GiB.  A step is one whole answer
    (expansion, conversion, 16 first-dimension sweeps, folding, packing, modulus switch).  One GPU: timed on the host around
    PackServer.answer (query upload and response download included: 64 KiB + 160 KiB against 11 ms).  N GPUs: the trials are
    independent up to the packing step, so the ranks split them (16 / N each); one all-gather of the 16 folded ciphertexts
    (512 KiB), rank 0 packs (include/spiral_gpu.h, spiral_gpu_pack_server_create_sharded)."""
    import numpy as np
    import torch

    import spiral_amd as sa

    world, rank, local_rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    out_n = 4
    pg = sa.make_params(10, 8, t_gsw=8, t_conv=4, t_exp=16, t_exp_right=56, qprime_bits=20, p_db=256)
    shp = sa.get_pack_shape(pg, out_n)
    if shp.trials % world:
        raise SystemExit(f"{shp.trials} trials do not split evenly over {world} ranks")
    use_dist = world > 1 or args.force_dist
    if args.shared_device:
        local_rank = 0
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if use_dist:
        import torch.distributed as dist

        kw = {"device_id": dev} if args.backend == "nccl" else {}
        dist.init_process_group(args.backend, rank=rank, world_size=world, **kw)
        # what the communicator actually saw, as in the base path's line
        me = {"rank": dist.get_rank(), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": torch.cuda.current_device(), "name": torch.cuda.get_device_name(dev), "pid": os.getpid()}
        seen = [None] * dist.get_world_size()
        dist.all_gather_object(seen, me)
        rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_seen": seen, "distinct_devices": len({r["device"] for r in seen}),
                "nccl_version": list(torch.cuda.nccl.version()) if args.backend == "nccl" else None}
    per = shp.trials // world
    srv = sa.PackServer(pg, out_n, local_rank, rank * per, (rank + 1) * per)
    stream = torch.cuda.Stream(device=dev)
    srv.set_stream(stream.cuda_stream)
    srv.gen_db(2024)
    rng = np.random.default_rng(1)  # the same synthetic keys and query on every rank
    srv.set_pub_params(synth_residues(rng, np, (shp.n_left, 2, pg.t_exp)), synth_residues(rng, np, (shp.n_right, 2, pg.t_exp_right)),
                       synth_residues(rng, np, (2, 2 * pg.t_conv)), synth_residues(rng, np, (out_n, out_n + 1, pg.t_conv)))
    q = synth_residues(rng, np, (shp.n_query_cts, 2))
    mine = torch.zeros(per * 2 * sa.N, dtype=torch.int64, device=dev)
    gathered = torch.zeros(shp.trials * 2 * sa.N, dtype=torch.int64, device=dev)
    us = []
    coll_ev = []  # HIP events around the collective, every step (one pair per step against an 11 ms step)

    def step():
        if not use_dist:
            us.append(srv.answer(q, want_packed=False)[2])
            return
        srv.fold_trials(q, mine.data_ptr())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        dist.all_gather_into_tensor(gathered, mine)  # 32 KiB per trial
        e1.record(stream)
        coll_ev.append((e0, e1))
        if rank == 0:
            srv.pack_gathered(gathered.data_ptr())
        us.append(srv.stage_us())

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    with torch.cuda.stream(stream):
        for _ in range(args.warmup):
            step()
        us.clear()
        coll_ev.clear()
        fence()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        fence()
        dt = time.perf_counter() - t0
        if use_dist:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
    ms = dt * 1e3 / args.steps
    sweep_ms = sum(u["sweep_kernels_us"] for u in us) / len(us) / 1e3
    nbytes = per * srv.sweep_bytes()
    achieved = nbytes / (sweep_ms * 1e-3) / 1e9
    out = {"metric": "server ms/query + DB GB/s vs HBM roofline (pack)", "value": round(ms, 4), "unit": "ms/query", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "ms_per_step": round(ms, 4), "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
           "dtype": "u32 residues, 32x32->64-bit integer MAC (two 28-bit CRT primes)", "data": "synthetic",
           "config": {"workload": "configs[4]: SpiralPack 2^18 x 30KB (nu1=10, nu2=8, n=4, p=256, t_GSW=8, t_conv=4, t_exp=16, q'=2^20), 16 explicit trial databases generated on device",
                      "db_bytes_ntt_form": int(shp.trials) * int(shp.dim0) * int(shp.num_per) * sa.N * 8,
                      "parallelism": f"trials x{world} ({per} per rank), one all-gather of the folded ciphertexts, rank 0 packs" if use_dist else "one GPU"},
           "stages_us": {k: round(sum(u[k] for u in us) / len(us), 1) for k in us[0]},
           "roofline": {"bound": "hbm", "kernel": f"sweep1_kernel ({per} trials per launch group, rank 0)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                        "traffic": None, "algorithmic_bytes_per_launch": int(nbytes), "avg_launch_ms": round(sweep_ms, 4)}}
    if use_dist:
        out["rccl"] = rccl
        out["collectives_us"] = {"all_gather_folded_trials": round(sum(a.elapsed_time(b) for a, b in coll_ev) / len(coll_ev) * 1e3, 1)}
    srv.close()
    if use_dist:
        dist.destroy_process_group()
    if rank == 0:
        import ctypes

        ctypes.CDLL(None).fflush(None)
        print(json.dumps(out), flush=True)


STREAM_ITEM_BYTES = 100_000  # BASELINE.json configs[3]: 2^20 x 100 KB


def bench_stream(args, ctx, prog, steps=None, warmup=None):
    """--workload stream: BASELINE.json configs[3] WHOLE.  A 100 KB item is factor = ceil(100000 / 15360) = 7 plaintexts of the "Streaming 20/spiralstream" set
    (n^2 * 2048 * log2(p) / 8 = 15 360 bytes each, select_params.py:297-298), i.e. SEVEN instances of the 2^20-item database, 64 GiB each in the reference's layout
    (56 GiB on the device): 448 GiB.  The client sends one query; the server converts it once and answers it against every instance -- first dimension, folding
    and response switch per instance, seven responses (the reference times one instance and multiplies fdim_us, fold_us and the response size by the factor,
    select_params.py:409-418).  The instances are independent: rank r holds instances r, r + N, ... (spiral_amd/dist.py instances_of_rank), every rank runs
    the conversion, ONE all-gather of the responses, no reduce.  A step = the whole item query on this rank's instances (spiral_gpu_server_run_query_instances,
    one hipGraph); value = wall ms per item query, max over ranks.  One GPU holds four instances (4 x 56 GiB + query state of 288 GiB): a rank that is assigned more
    instances than fit re-sweeps its resident images for the rest -- the timing of the sweep does not depend on the data -- and says so (instances_resident)."""
    import hashlib

    inject_fail("stream_item", ctx.rank)

    import numpy as np

    import spiral_amd as sa
    from spiral_amd import dist as sdist
    from spiral_amd import server as SV

    torch, dist = ctx.torch, ctx.dist
    world, rank, dev = ctx.world, ctx.rank, ctx.dev
    params_kw = dict(WORKLOADS["stream"])
    label = params_kw.pop("label")
    if args.nu1 is not None: params_kw["nu1"] = args.nu1  # (self-tests: a geometry whose seven instances fit one GPU)
    if args.nu2 is not None: params_kw["nu2"] = args.nu2
    if (args.nu1, args.nu2) != (None, None): label += f" with nu1={params_kw['nu1']}, nu2={params_kw['nu2']}"
    pg = sa.make_params(**params_kw)
    shp = sa.get_shape(pg)
    plain_bytes = 4 * sa.N * 15 // 8  # n0 * n2 * 2048 coefficients of log2(p) = 15 bits
    factor = -(-STREAM_ITEM_BYTES // plain_bytes)
    mine = sdist.instances_of_rank(rank, world, factor)
    slots = -(-factor // world)
    prog.arm("stream/set-up", 6.0)
    image_bytes = sa.N * shp.dim0 * shp.num_per * 4 * 7  # 3.5 bytes per residue
    free, _total = torch.cuda.mem_get_info(dev)
    if args.shared_device: free //= world  # (self-test: the ranks share one device)
    fit = max(1, int((free - (8 << 30)) // (image_bytes + (4 << 30))))  # an image + a server's query state (~3 GiB at these parameters), 8 GiB of headroom
    stream = torch.cuda.Stream(device=dev)
    inst = []
    for k in mine[:fit]:
        sv = sa.Server(pg, ctx.local_rank)
        sv.set_stream(stream.cuda_stream)
        sv.gen_db(DB_SEED + k)  # instance k = its own database
        inst.append(sv)
    resident = len(inst)
    swept = [inst[i % resident] for i in range(len(mine))] if mine else []  # (instances beyond the resident ones re-sweep resident images)
    qsrv = inst[0] if inst else None
    steps, warmup = steps or args.steps, args.warmup if warmup is None else warmup
    resp = torch.zeros(slots * 6 * sa.N, dtype=torch.int64, device=dev)
    gathered = torch.zeros(world * resp.numel(), dtype=torch.int64, device=dev) if ctx.use_dist else resp
    if qsrv is not None:
        pub, query = synth_inputs(np, sa, pg, shp)
        qsrv.set_pub_params(*pub)
        qsrv.set_query(query)
        qsrv.use_graphs(not args.no_graphs)

    def step():
        if qsrv is not None:
            qsrv.run_query_instances(swept, resp.data_ptr(), pre=True)
        if ctx.use_dist:
            sdist.all_gather_instance_responses(gathered, resp)

    with torch.cuda.stream(stream):
        prog.arm("stream/timed")
        step()  # graph capture, untimed
        for _ in range(warmup):
            step()
        ctx.fence()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        ctx.fence()
        ms = ctx.max_over_ranks(time.perf_counter() - t0) * 1e3 / steps
        prog.arm("stream/stage split", 3.0)
        out_h = None
        if rank == 0:
            h = hashlib.sha256()
            for k in range(factor):
                h.update(sdist.instance_response(gathered, k, world, slots).cpu().numpy().astype("<i8").tobytes())
            out_h = h.hexdigest()
        # stage split on rank 0's first instance, outside the timed region: conversion alone, one instance's sweep + fold + switch, the sweep kernel alone
        pre_us = inst_us = sweep_ms = None
        if rank == 0 and qsrv is not None:
            e = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
            qsrv.run_pre(); qsrv.run_query_instances(swept[:1], resp.data_ptr(), pre=False)  # captures
            torch.cuda.synchronize()
            n = 5
            e[0].record(stream)
            for _ in range(n): qsrv.run_pre()
            e[1].record(stream)
            for _ in range(n): qsrv.run_query_instances(swept[:1], resp.data_ptr(), pre=False)
            e[2].record(stream)
            torch.cuda.synchronize()
            pre_us, inst_us = e[0].elapsed_time(e[1]) * 1e3 / n, e[1].elapsed_time(e[2]) * 1e3 / n
            sweep_ms = qsrv.time_sweep(5)
    line = None
    if rank == 0:
        bytes_sweep, dev_bytes = qsrv.sweep_bytes(), qsrv.sweep_device_bytes()
        achieved = bytes_sweep / (sweep_ms * 1e-3) / 1e9
        wire = int(sa.response_wire_bytes(pg))
        line = {
            "metric": "server ms/query + DB GB/s vs HBM roofline (stream: configs[3], whole 100 KB items)", "value": round(ms, 4), "unit": "ms/query", "n_gpus": world, "steps": steps, "warmup": warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": False, "scaling": "strong", "vs_baseline": None,
            "dtype": "u32 residues, 32x32->64-bit integer MAC (two 28-bit CRT primes)", "data": "synthetic",
            "config": {"workload": label + f"; an item = factor {factor} instances of the 2^20 x {plain_bytes} B database ({factor} x 64 GiB NTT form), explicit DBs generated on device",
                       "answer_sha256": out_h, "db_bytes_ntt_form": factor * int(shp.dim0) * int(shp.num_per) * 4 * sa.N * 8,
                       "parallelism": f"instances x{world} (rank r: instances r, r + {world}, ...), conversion on every rank, one all-gather of the {factor} responses, no reduce" if ctx.use_dist else
                                      f"one GPU: {factor} instances swept one after the other"},
            "item": {"factor": factor, "item_bytes": STREAM_ITEM_BYTES, "plaintext_bytes": plain_bytes, "instances_of_rank0": mine, "instances_resident": resident,
                     "instances_swept_per_query_rank0": len(swept), "image_bytes_per_instance": image_bytes, "db_device_bytes_rank0": int(sum(sv.db_device_bytes() for sv in inst)),
                     "conversion_us": round(pre_us, 1), "per_instance_us": round(inst_us, 1), "sweep_kernel_ms": round(sweep_ms, 4),
                     "item_ms_by_the_reference_formula": round((pre_us + factor * inst_us) / 1e3, 4), "response_bytes": factor * wire,
                     "note": "value = one timed item query on this many GPUs.  item_ms_by_the_reference_formula = conversion + factor x (first dimension + folding + switch) of ONE instance, "
                             "the way select_params.py:409-426 prices an item from a single-instance run.  instances_resident < instances_swept: the rank re-sweeps resident images for "
                             "the instances that do not fit (one GPU holds four 56 GiB images), the sweep's time does not depend on the data"},
            "answer_sha256": out_h,
            "roofline": {"bound": "hbm", "kernel": "sweep_kernel (one instance)", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBPS, 4),
                         "frac_device_bytes": round(dev_bytes / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": None, "algorithmic_bytes_per_launch": int(bytes_sweep),
                         "device_bytes_per_launch": int(dev_bytes), "avg_launch_ms": round(sweep_ms, 4), "launches_timed": 5,
                         "database_bytes_per_s_over_the_item_query": round(len(swept) * bytes_sweep / (ms * 1e-3) / 1e9, 1)},
        }
        if ctx.use_dist:
            line["rccl"] = ctx.rccl
    for sv in inst:
        sv.close()
    torch.cuda.empty_cache()
    return line


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="config2", choices=sorted(WORKLOADS) + ["pack", "stream-instance"], help="config2 = BASELINE.json configs[1], the one the metric is quoted on; "
                    "config3 / stream / pack = configs[2] / [3] / [4]; stream = a whole 100 KB item (7 database instances, factor-sharded over the GPUs), "
                    "stream-instance = one of its instances through the j-shard path (the figure of rounds 1-5)")
    ap.add_argument("--headline", default=None, choices=["config2", "config3"], help="which of the two base geometries is `value` when both are timed (default config2 = configs[1], "
                    "the one the metric is quoted on, with configs[2]'s geometry under `also.config3`); --headline config3 swaps them: `value` is then the 2^24 x 256 B geometry, "
                    "whose sweep is 75 %% of the query -- the one a j-shard over N GPUs CAN scale -- and configs[1] goes under `also.config2`")
    ap.add_argument("--pg-watchdog", type=float, default=float(os.environ.get("SPIRAL_BENCH_PG_WATCHDOG_S", "90")), help="seconds the process-group set-up and its collective smoke "
                    "test may take (N > 1) before the run exits 3 with a line that says which ranks never joined")
    ap.add_argument("--nu1", type=int, default=None, help="override the workload's first-dimension size (tuning)")
    ap.add_argument("--nu2", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--watchdog", type=float, default=float(os.environ.get("SPIRAL_BENCH_WATCHDOG_S", "180")), help="seconds a schedule / phase may take before the line-so-far "
                    "is printed (partial: true, hung_in) and the process exits 3; 0 = off")
    ap.add_argument("--prewarm", type=int, default=40, help="untimed queries run as part of the set-up before the W warm-up steps, to ramp the GPU's clocks (0 = none)")
    ap.add_argument("--no-graphs", action="store_true", help="launch every kernel eagerly instead of replaying hipGraphs")
    ap.add_argument("--overlap", type=int, default=0, choices=[0, 2], help="2: the split schedule -- the odd tree of the expansion and the Regev->GSW "
                    "conversion as their own launch sequence on a side stream beside the even tree + ScalToMat + sweep; 0: everything in order on one stream")
    ap.add_argument("--event-every", type=int, default=5, help="bracket the stages with HIP events on every n-th timed step only (1 = every step)")
    ap.add_argument("--root-fold", action="store_true", help="N > 1: plain reduce to rank 0, which lifts and folds alone")
    ap.add_argument("--no-batched-sweep", dest="batched_sweep", action="store_false", help="N = 1: skip the batched-sweep part of the throughput leg")
    ap.add_argument("--lanes", type=int, default=3, help="N = 1: queries in flight in the extra throughput leg (`pipelined` in the JSON line; 1 = skip it)")
    ap.add_argument("--schedule", default="all", choices=["all", "both", "in-order", "comm-overlap", "pipelined"], help="N > 1 with the sharded expansion and the distributed fold: "
                    "in-order = every collective where its result is needed; comm-overlap = the all-gather of the GSW bits under ScalToMat + sweep and the "
                    "Regev->GSW conversion under the reduce-scatter (async collectives); pipelined = comm-overlap with the sweep issued in --sweep-stages column-block "
                    "stages, each stage's accumulators reduce-scattered while the next stage sweeps; all (default) = each timed over the K steps: `value` is the "
                    "comm-overlap one (a fixed schedule, so that rounds compare like with like) and `schedules` holds all of them")
    ap.add_argument("--sweep-stages", type=int, default=4, help="stages of the pipelined schedule (clamped to what the geometry allows: whole 64-column blocks per stage)")
    ap.add_argument("--comm-overlap", action="store_true", help="same as --schedule comm-overlap")
    ap.add_argument("--replicated-expansion", action="store_true", help="N > 1: every rank runs the whole query expansion (default: each rank expands its own "
                    "first-dimension subtree and every N-th GSW bit, one all-gather of the GSW bits)")
    ap.add_argument("--no-config3", action="store_true", help="skip the secondary leg (`also.config3`: BASELINE.json configs[2]'s 2^24 x 256 B geometry, the one whose "
                    "sweep is most of the query, timed after the headline workload in the same invocation so that the N = 1, 2, 4, 8 runs give its curve too)")
    ap.add_argument("--config3-steps", type=int, default=10)
    ap.add_argument("--no-stream-item", action="store_true", help="skip the last leg (`also.stream_item`: BASELINE.json configs[3] whole -- a 100 KB item = 7 instances of the 64 GiB "
                    "database, factor-sharded over the ranks, one all-gather and no reduce: the configuration whose N-GPU scaling is near-linear up to 7)")
    ap.add_argument("--no-replicas", action="store_true", help="N > 1: skip the `replicas` block (every rank answering batches of whole queries on its own full copy of the database)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only for the one-GPU self-test)")
    ap.add_argument("--shared-device", action="store_true", help="self-test: all ranks use device 0")
    ap.add_argument("--force-dist", action="store_true", help="initialise torch.distributed and run the reduce even with one rank (self-test)")
    args = ap.parse_args(argv)
    if args.comm_overlap:
        args.schedule = "comm-overlap"
    if args.headline == "config3" and args.workload == "config2":
        args.workload, args.secondary = "config3", "config2"
    else:
        args.secondary = "config3" if args.workload == "config2" else None
    return args


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher in the environment: start the N ranks as fresh child processes
    (torch.distributed.run, one per GPU, rendezvous on 127.0.0.1) BEFORE this process imports torch or touches a GPU, pass
    their output through (rank 0 prints the JSON line) and return their exit code.  Nothing is exec'd from a process that has
    initialised the GPU: this parent never does."""
    import subprocess

    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this driver
    # --standalone: the launcher's own c10d rendezvous on a port it binds itself (no bind-close-reuse race with another job on the box)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--standalone", "--local-addr", "127.0.0.1", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           os.path.abspath(__file__)] + list(argv)
    print(f"bench.py: no WORLD_SIZE in the environment, launching {args.gpus} ranks: {' '.join(cmd)}", file=sys.stderr, flush=True)
    return subprocess.call(cmd, env=env, cwd=ROOT)


def _pg_checkin(rank):
    """a file per rank in a directory named after the rendezvous: who has reached the process-group set-up (one node, so /tmp is shared)"""
    import atexit
    import tempfile

    d = os.path.join(tempfile.gettempdir(), "spiral_bench_pg_%s_%s" % (os.environ.get("MASTER_PORT", "0"), os.environ.get("TORCHELASTIC_RUN_ID", "none")))
    os.makedirs(d, exist_ok=True)
    f = os.path.join(d, f"rank{rank}")
    with open(f, "w") as fh:
        fh.write(str(time.time()))
    atexit.register(lambda: os.path.exists(f) and os.remove(f))
    return d


def _pg_absent(d, world):
    """for the watchdog's line: the ranks whose check-in file is missing (or older than ten minutes: a previous run's)"""
    now, present = time.time(), []
    for r in range(world):
        try:
            if now - os.path.getmtime(os.path.join(d, f"rank{r}")) < 600:
                present.append(r)
        except OSError:
            pass
    return {"ranks_checked_in": present, "ranks_never_joined": [r for r in range(world) if r not in present]}


class Ctx:
    """what every leg of one invocation shares: the rank layout and the (once-initialised) process group"""

    def __init__(self, args, prog=None):
        import torch

        self.torch = torch
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.rank = int(os.environ.get("RANK", "0"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", "0"))
        if self.world != args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={self.world}")
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
        if args.shared_device:  # self-test of the N > 1 flow on a one-GPU box: every rank on device 0 (use with --backend gloo)
            self.local_rank = 0
        torch.cuda.set_device(self.local_rank)
        self.dev = torch.device("cuda", self.local_rank)
        self.use_dist = self.world > 1 or args.force_dist
        self.dist = None
        self.rccl = None
        if self.use_dist:
            import torch.distributed as dist

            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", "29517")
            os.environ.setdefault("RANK", "0")
            os.environ.setdefault("WORLD_SIZE", "1")
            # The first real multi-rank run must say in seconds, not minutes, what went wrong: every rank checks in with a file before the (blocking)
            # set-up, and the set-up + a collective smoke test run under their own short watchdog whose line names the ranks that never arrived.
            here = _pg_checkin(self.rank)
            if prog is not None:
                prog.on_hang = lambda: _pg_absent(here, self.world)
                prog.arm("process group set-up", seconds=args.pg_watchdog)
            if os.environ.get("SPIRAL_BENCH_INJECT_HANG", "") == f"pg-setup:{self.rank}":  # test hook: this rank never joins
                print(f"bench.py: SPIRAL_BENCH_INJECT_HANG: rank {self.rank} never joins the process group", file=sys.stderr, flush=True)
                os.remove(os.path.join(here, f"rank{self.rank}"))
                time.sleep(10 ** 6)
            if args.backend == "nccl":
                dist.init_process_group(backend="nccl", device_id=self.dev)
            else:
                dist.init_process_group(backend=args.backend)
            self.dist = dist
            # what the communicator actually saw (not what the command line asked for): every rank's device, gathered
            mine = {"rank": dist.get_rank(), "local_rank": int(os.environ.get("LOCAL_RANK", "0")), "device": torch.cuda.current_device(),
                    "name": torch.cuda.get_device_name(self.dev), "pid": os.getpid()}
            seen = [None] * dist.get_world_size()
            dist.all_gather_object(seen, mine)
            self.rccl = {"backend": dist.get_backend(), "world_size": dist.get_world_size(), "ranks_seen": seen,
                         "distinct_devices": len({(r["device"]) for r in seen}), "nccl_version": list(torch.cuda.nccl.version()) if args.backend == "nccl" else None}
            # collective smoke test, still under the short watchdog: the three collectives of the data path on 4-byte-per-rank tensors, checked.  An
            # xGMI / IPC fault shows up here, with ranks_seen already in the line, not minutes later inside a timed schedule.
            if prog is not None:
                prog.line = {"metric": "server ms/query + DB GB/s vs HBM roofline", "value": None, "rccl": self.rccl}
                prog.arm("collective smoke test (all_reduce, reduce_scatter_tensor, all_gather_into_tensor)", seconds=args.pg_watchdog)
            w, r = dist.get_world_size(), dist.get_rank()
            t0 = time.perf_counter()
            a = torch.full((1,), r + 1, dtype=torch.int32, device=self.dev)
            dist.all_reduce(a)
            src = torch.arange(w, dtype=torch.int32, device=self.dev) + r
            part = torch.zeros(1, dtype=torch.int32, device=self.dev)
            if args.backend == "gloo":  # (the one-GPU self-test: gloo has no reduce-scatter, spiral_amd/dist.py emulates it the same way)
                tmp = src.clone()
                dist.all_reduce(tmp)
                part.copy_(tmp[r:r + 1])
            else:
                dist.reduce_scatter_tensor(part, src)
            allg = torch.zeros(w, dtype=torch.int32, device=self.dev)
            dist.all_gather_into_tensor(allg, part)
            torch.cuda.synchronize()
            want = [k * w + w * (w - 1) // 2 for k in range(w)]  # rank k's reduce-scattered word: sum over ranks r of (k + r)
            ok = int(a.item()) == w * (w + 1) // 2 and allg.tolist() == want
            self.rccl["collective_smoke"] = {"ok": bool(ok), "ms": round((time.perf_counter() - t0) * 1e3, 1)}
            if not ok:
                raise SystemExit(f"bench.py: rank {r}: the collective smoke test gave wrong sums (all_reduce {int(a.item())}, all_gather {allg.tolist()}, expected {want})")
            if prog is not None:
                prog.on_hang = None

    def fence(self):
        if self.use_dist:
            self.dist.barrier()
        self.torch.cuda.synchronize()

    def max_over_ranks(self, dt):
        if not self.use_dist:
            return dt
        t = self.torch.tensor([dt], dtype=self.torch.float64, device=self.dev)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def close(self):
        if self.use_dist:
            self.dist.destroy_process_group()


def bench_base(args, ctx, workload, steps, warmup, primary, prog):
    """one leg: `steps` timed queries of `workload` through the base server on ctx.world ranks.  primary: the headline leg, which
    also carries the throughput leg, the transform roofline and the reference-bucket detail.  prog: the line-so-far + watchdog."""
    import numpy as np

    import spiral_amd as sa
    from spiral_amd import dist as sdist
    from spiral_amd import server as SV

    torch, dist = ctx.torch, ctx.dist
    world, rank, local_rank, dev, use_dist = ctx.world, ctx.rank, ctx.local_rank, ctx.dev, ctx.use_dist
    params_kw = dict(WORKLOADS[workload])
    label = params_kw.pop("label")
    if primary:
        if args.nu1 is not None: params_kw["nu1"] = args.nu1
        if args.nu2 is not None: params_kw["nu2"] = args.nu2
        if (args.nu1, args.nu2) != (None, None): label += f" with nu1={params_kw['nu1']}, nu2={params_kw['nu2']}"
    nu1, nu2 = params_kw["nu1"], params_kw["nu2"]
    pg = sa.make_params(**params_kw)
    shp = sa.get_shape(pg)
    j0, j1 = sdist.shard_range(rank, world, shp.dim0)

    prog.arm(f"{workload}/set-up", 3.0)
    srv = sa.Server(pg, local_rank, j0, j1)
    stream = torch.cuda.Stream(device=dev)  # a real (capturable) stream shared by the library and RCCL
    srv.set_stream(stream.cuda_stream)
    srv.gen_db(DB_SEED)  # explicit database generated on the device, this rank's j-shard
    pub, query = synth_inputs(np, sa, pg, shp)  # same synthetic inputs on every rank
    srv.set_pub_params(*pub)
    srv.set_query(query)
    acc = torch.zeros(shp.num_per * 6 * sa.N, dtype=torch.int64, device=dev)
    srv.set_acc(acc.data_ptr())
    # distributed fold (reduce-scatter variant): the ranks' accumulators are reduce-scattered by ciphertext,
    # every rank lifts + folds its num_per/G ciphertexts, the G survivors are all-gathered and rank 0 finishes
    G = sdist.fold_ranks(world, shp.num_per) if use_dist and not args.root_fold else 1
    if G > 1 or (use_dist and world == 1 and not args.root_fold):
        srv.set_fold_ranks(G)
        chunk = torch.zeros(acc.numel() // G, dtype=torch.int64, device=dev)
        ct = torch.zeros(6 * sa.N, dtype=torch.int64, device=dev)
        gathered = torch.zeros(G * 6 * sa.N, dtype=torch.int64, device=dev)
        sharded_fold = True
    else:
        sharded_fold = False
    # sharded expansion: the expansion is database-independent, so N ranks would each repeat all of it; instead a rank expands the
    # subtree above its own j-block and every N-th GSW bit, and the GSW bits are all-gathered
    shard_expand = use_dist and not args.replicated_expansion and sdist.expand_shard_ok(shp, pg, world)
    if shard_expand:
        srv.set_expand_shard(rank, world)
        bits = torch.zeros(srv.gsw_bits_words(), dtype=torch.int64, device=dev)
        bits_all = torch.zeros(world * bits.numel(), dtype=torch.int64, device=dev)
    can_overlap = shard_expand and sharded_fold
    n_stages = min(max(1, args.sweep_stages), srv.max_sweep_stages()) if can_overlap else 1
    while n_stages & (n_stages - 1): n_stages -= 1
    # the schedules are timed safest first (in-order, comm-overlap, pipelined): a failure in a later one leaves the earlier numbers in the line
    if not can_overlap:
        schedules = ["in-order"]
    elif args.schedule in ("all", "both"):  # at world size 1 there is nothing to hide: the comparison is for real multi-rank runs (and the self-tests)
        schedules = ["in-order", "comm-overlap"] if (world > 1 or args.force_dist) else ["in-order"]
        if args.schedule == "all" and len(schedules) > 1 and n_stages > 1: schedules.append("pipelined")
    else:
        schedules = [args.schedule]
    if "pipelined" in schedules and n_stages < 2:
        raise SystemExit("--schedule pipelined: this geometry does not split into sweep stages (needs num_per >= 64)")
    srv.use_graphs(not args.no_graphs)  # expand+convert and lift+fold+finish replay as two hipGraphs
    srv.set_overlap(int(args.overlap))

    # HIP events bracket the stages (and give the sweep's launch duration for the roofline) on every event_every-th timed
    # step; an event record costs the stream ~6 us, so the other steps replay the whole query (one GPU) or everything before
    # the collective (N GPUs) as one graph.  On those steps every collective is bracketed too (`collectives_us`).
    sampled = [k for k in range(steps) if k % max(1, args.event_every) == 0]
    new_events = lambda: [torch.cuda.Event(enable_timing=True) for _ in range(10)]
    ev = {k: new_events() for k in sampled}

    whole = world == 1 and not use_dist and not args.no_graphs and not args.overlap and args.event_every > 1

    cur_stages = [1]

    def configure(sch):
        # the pipelined schedule lays the accumulators out [stage][rank][ct]; the others need the single-stage layout (one reduce-scatter).
        # (changing the layout drops the captured graphs, so it is only touched when the stage count really changes)
        want = n_stages if sch == "pipelined" else 1
        if can_overlap and n_stages > 1 and want != cur_stages[0]:
            srv.set_sweep_stages(want)
            cur_stages[0] = want

    inject_hang = os.environ.get("SPIRAL_BENCH_INJECT_HANG", "")  # test hook: the named schedule (or "<workload>/<schedule>") never returns

    def step(e=None, overlap_comm=False, staged=False):
        # one query: [expand, convert] -> sweep -> [reduce over ranks] -> [lift, fold, response switch]
        if e is None and whole:
            srv.run_query()  # the same kernels as below, replayed as ONE hipGraph: no event / launch seams around the sweep
            return
        if e:
            e[0].record(stream)
            if shard_expand:
                srv.run_expand_pack(bits.data_ptr())
                e[4].record(stream)
                sdist.all_gather_gsw_bits(bits_all, bits)
                e[5].record(stream)
                srv.gsw_bits_unpack(bits_all.data_ptr())
                srv.convert()
            else:
                srv.run_pre()
            e[1].record(stream)
            srv.first_dim()
            e[2].record(stream)
        elif shard_expand and staged:
            # comm-overlap, with the sweep in n_stages column-block launches: stage k's accumulators (a contiguous 1/n_stages of the buffer)
            # are reduce-scattered while stage k + 1 streams the database
            srv.run_expand_pack(bits.data_ptr())
            w_bits = sdist.all_gather_gsw_bits(bits_all, bits, async_op=True)
            srv.run_scal2mat()
            works = sdist.reduce_scatter_stages(chunk, acc, n_stages, async_op=True, after_stage=srv.first_dim_stage)
            w_bits.wait()
            srv.run_unpack_gsw(bits_all.data_ptr())
            for wk in works:
                if wk is not None: wk.wait()
            srv.fold_local(chunk.data_ptr(), ct.data_ptr())
            sdist.all_gather_cts(gathered, ct)
            if rank == 0:
                srv.fold_root(gathered.data_ptr())
            return
        elif shard_expand and overlap_comm:
            # the GSW bits only feed the folding keys: their all-gather runs under ScalToMat + sweep, and the Regev->GSW conversion
            # that consumes them under the reduce-scatter of the accumulators
            srv.run_expand_pack(bits.data_ptr())
            w_bits = sdist.all_gather_gsw_bits(bits_all, bits, async_op=True)
            srv.run_scal2mat_sweep()
            w_acc = sdist.reduce_scatter_accumulators(chunk, acc, async_op=True)
            w_bits.wait()
            srv.run_unpack_gsw(bits_all.data_ptr())
            if w_acc is not None: w_acc.wait()
            srv.fold_local(chunk.data_ptr(), ct.data_ptr())
            sdist.all_gather_cts(gathered, ct)
            if rank == 0:
                srv.fold_root(gathered.data_ptr())
            return
        elif shard_expand:
            srv.run_expand_pack(bits.data_ptr())       # this rank's share of the expansion, one graph
            sdist.all_gather_gsw_bits(bits_all, bits)  # every rank needs every GSW bit
            srv.run_unpack_convert_sweep(bits_all.data_ptr())
        else:
            srv.run_pre_sweep()  # one graph for everything before the collective
        if sharded_fold:
            if e: e[6].record(stream)
            sdist.reduce_scatter_accumulators(chunk, acc)
            if e: e[7].record(stream)
            srv.fold_local(chunk.data_ptr(), ct.data_ptr())
            if e: e[8].record(stream)
            sdist.all_gather_cts(gathered, ct)
            if e: e[9].record(stream)
            if rank == 0:
                srv.fold_root(gathered.data_ptr())
        else:
            if use_dist:
                if e: e[6].record(stream)
                sdist.reduce_accumulators(acc, dst=0)
                if e: e[7].record(stream)
            if rank == 0:
                srv.run_post(reduce_first=use_dist)
        if e: e[3].record(stream)

    def timed_loop(sch, evs):
        """W warm-up steps, then exactly `steps` timed steps between two fences; ms per step, max over ranks.  evs: the events of the sampled steps"""
        ov, st = sch == "comm-overlap", sch == "pipelined"
        for i in range(warmup):
            step(None, ov, st)
        ctx.fence()
        t0 = time.perf_counter()
        for k in range(steps):
            # the stage events belong to the in-order flavour (with the others the stages interleave); every schedule still runs exactly `steps` steps
            step(evs[k] if (evs and k in sampled) else None, ov, st)
        ctx.fence()
        return ctx.max_over_ranks(time.perf_counter() - t0) * 1e3 / steps

    timed, hashes, no_prewarm, ev_done = {}, {}, None, [False]
    names = ["expand_convert", "sweep", "reduce_lift_fold_switch"]  # the last bucket includes the collective(s)

    def make_out(extra=None):
        """the JSON line from what has been measured so far (called after every schedule, and at the end with everything else)"""
        # N > 1: the headline is ONE fixed schedule (comm-overlap where it applies), not the best of several -- a minimum over noisy
        # timings is biased low and would not compare like with like between rounds; the others are reported beside it in `schedules`
        fastest = min(timed, key=timed.get)
        best = "comm-overlap" if "comm-overlap" in timed else ("in-order" if "in-order" in timed else fastest)
        ms_per_step = timed[best]
        have_ev = ev_done[0] and bool(sampled)  # the stage events have been recorded (an in-order pass ran)
        stages = {n: sum(ev[k][i].elapsed_time(ev[k][i + 1]) for k in sampled) / len(sampled) * 1e3 for i, n in enumerate(names)} if have_ev else None
        coll = {}
        if use_dist and have_ev:
            pairs = ([("all_gather_gsw_bits", 4, 5)] if shard_expand else []) + ([("reduce_scatter_accumulators", 6, 7), ("all_gather_folded_cts", 8, 9)] if sharded_fold else [("reduce_accumulators", 6, 7)])
            coll = {n: round(sum(ev[k][a].elapsed_time(ev[k][b]) for k in sampled) / len(sampled) * 1e3, 1) for n, a, b in pairs}
        bytes_sweep = srv.sweep_bytes()
        traffic, traffic_src = pmc_traffic(world, nu1, nu2)
        out = {
            "metric": "server ms/query + DB GB/s vs HBM roofline" + (", 2^20 x 256B" if workload == "config2" else f" ({workload})"),
            "value": round(ms_per_step, 4),
            "unit": "ms/query",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": False,
            "scaling": "strong",
            "vs_baseline": None,
            "dtype": "u32 residues, 32x32->64-bit integer MAC (two 28-bit CRT primes)",
            "data": "synthetic",
            "config": {"workload": label + ", explicit DB generated on device, sharded by first-dimension index",
                       # (inside `config` because the driver's record keeps this dict whole: the hash of what the timed graph computed -- the full-size GPU
                       # test prints the oracle's hash for the same inputs -- and the no-pre-warm protocol's value)
                       "answer_sha256": hashes.get(best) or (next(iter(hashes.values())) if hashes else None),
                       "value_no_prewarm": no_prewarm["value"] if no_prewarm else None,
                       "db_bytes_ntt_form": int(shp.dim0) * int(shp.num_per) * 4 * sa.N * 8, "schedule": "split: the GSW side of the query on a side stream" if args.overlap else "in order, one stream",
                       "parallelism": (f"j-shard x{world}, reduce-scatter + distributed fold + all-gather" if sharded_fold else f"j-shard x{world} + 1 reduce")
                                      + (", sharded expansion + all-gather of the GSW bits" if shard_expand else "")
                                      + (" (overlapped with ScalToMat + sweep)" if best == "comm-overlap" else "")
                                      + (f" (sweep pipelined with its reduce-scatter in {n_stages} stages)" if best == "pipelined" else "")},
            "queries_per_s": round(1e3 / ms_per_step, 2),
            "prewarm_queries": args.prewarm,
            "value_no_prewarm": no_prewarm,
            "answer_sha256": hashes.get(best) or (next(iter(hashes.values())) if hashes else None),
            "stages_us": {k: round(v, 1) for k, v in stages.items()} if stages else None,
        }
        if stages:
            sweep_ms = stages["sweep"] / 1e3
            achieved = bytes_sweep / (sweep_ms * 1e-3) / 1e9
            out["roofline"] = {"bound": "hbm", "kernel": "sweep_kernel", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBPS, 4), "frac_device_bytes": round(srv.sweep_device_bytes() / (sweep_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": int(bytes_sweep),
                               "avg_launch_ms": round(sweep_ms, 4),
                               "device_bytes_per_launch": int(srv.sweep_device_bytes()), "achieved_device_bytes": round(srv.sweep_device_bytes() / (sweep_ms * 1e-3) / 1e9, 1),
                               "note": "achieved / frac = SURVEY 8d algorithmic bytes (8 B per database word) / launch time measured in THIS run with HIP events on the launch stream; "
                                       "the device keeps a word's two 28-bit residues in 7 bytes, so a launch physically moves device_bytes_per_launch: achieved_device_bytes / "
                                       "frac_device_bytes are the HBM utilisation in physical bytes.  `traffic` is NOT measured in this run: it is the rocprofv3 PMC figure "
                                       "(FETCH_SIZE / WRITE_SIZE passes) read from the committed file named in traffic_source, valid for configs[1] on one GPU only (null otherwise)",
                               "launches_timed": len(sampled), "shard": f"j in [{j0},{j1}) on rank 0"}
        if use_dist:
            out["schedules"] = {"ms_per_query": {k: round(v, 4) for k, v in timed.items()}, "chosen": best, "fastest": fastest, "sweep_stages": n_stages if "pipelined" in timed else None,
                                "answer_sha256": dict(hashes), "requested": list(schedules),
                                "note": "each schedule timed over the same K steps after W warm-up steps, max over ranks, safest first (a later schedule that fails leaves the earlier ones in the line); "
                                        "value = the `chosen` one, fixed in advance (comm-overlap where the sharded expansion and the distributed fold apply), `fastest` names the minimum; "
                                        "answer_sha256: rank 0's answer after each schedule's last step (equal: the schedules compute the same function)"}
            out["collectives_us"] = coll
            out["rccl"] = ctx.rccl
        if extra:
            for k, v in extra.items():
                if isinstance(v, dict) and isinstance(out.get(k), dict):
                    out[k].update(v)
                else:
                    out[k] = v
        return out

    def read_answer():
        """rank 0: sha256 of the folded ciphertext + response the last step left (the stream is idle: every caller fenced)"""
        return answer_hash(np, srv.read(SV.BUF_FINAL), srv.read(SV.BUF_RESPONSE)) if rank == 0 else None

    with torch.cuda.stream(stream):
        for n_sch, sch in enumerate(schedules):
            prog.arm(f"{workload}/{sch}")
            ov, st = sch == "comm-overlap", sch == "pipelined"
            configure(sch)
            if not args.no_graphs:  # priming, not a step of the run: each step flavour once, so that no hipGraph is captured in the timed region
                if sch == "in-order": step(new_events())
                step(None, ov, st)
            inject_fail(sch, rank)
            if inject_hang in (sch, f"{workload}/{sch}"):
                print(f"bench.py: SPIRAL_BENCH_INJECT_HANG={inject_hang}: rank {rank} stops here", file=sys.stderr, flush=True)
                time.sleep(10 ** 6)
            if n_sch == 0 and args.prewarm > 0:
                # the round-3 protocol, for comparison across rounds: W warm-up + K timed steps straight after the set-up, no clock pre-warm
                no_prewarm = {"value": round(timed_loop(sch, {k: new_events() for k in sampled} if sch == "in-order" else None), 4), "unit": "ms/query", "schedule": sch,
                              "note": "the same K timed steps after W warm-up steps but BEFORE the pre-warm queries (the protocol of rounds 1-3): the GPU's clocks are still ramping"}
            # clock pre-warm, part of the set-up like the priming above (reported as `prewarm_queries`): the GPU's clocks ramp over the first tens of
            # milliseconds of work after the host-side set-up, and the latency-bound stages of the first ~30 queries run 3-5 % slower than the steady
            # state this benchmark is about (profiles/r04_bench_warmup.txt); the W warm-up steps and the K timed steps follow unchanged
            for i in range(args.prewarm):
                step(None, ov, st)
            timed[sch] = timed_loop(sch, ev if sch == "in-order" else None)
            if sch == "in-order": ev_done[0] = True
            hashes[sch] = read_answer()
            prog.update(make_out())
        prog.arm(f"{workload}/after the timed schedules", 3.0)
        if "in-order" not in timed:  # a single other schedule was asked for: the stage split still comes from a few in-order steps, outside the timed region
            configure("in-order")
            step(None)  # untimed, unsampled: the graphs the layout change dropped are re-captured here, not inside a sampled step
            for k in sampled:
                step(ev[k])
            ctx.fence()
            ev_done[0] = True
        configure("in-order")
        # throughput leg (outside the timed region, reported beside `value`, never as it): `lanes` queries in flight on one database
        # image, one server handle and one stream per lane, each replaying the whole-query graph
        pipelined = None
        if whole and args.lanes > 1:
            n_lanes = max(args.lanes, 16 if args.batched_sweep else 0)
            lanes = [(srv, stream)]
            for _ in range(n_lanes - 1):
                lane, lane_stream = sa.Server(pg, local_rank, j0, j1, share_db_of=srv), torch.cuda.Stream(device=dev)
                lane.set_stream(lane_stream.cuda_stream)
                lane.set_pub_params(*pub)
                lane.set_query(query)
                lane.use_graphs(True)
                lanes.append((lane, lane_stream))
            for lane, _ in lanes:
                lane.run_query()  # graph capture, untimed
            torch.cuda.synchronize()
            n_q = max(steps, 100 if primary else 24) // args.lanes * args.lanes
            t1 = time.perf_counter()
            for k in range(n_q):
                lanes[k % args.lanes][0].run_query()
            torch.cuda.synchronize()
            dt_p = time.perf_counter() - t1
            pipelined = {"lanes": args.lanes, "queries": n_q, "queries_per_s": round(n_q / dt_p, 1), "ms_per_query_amortised": round(dt_p * 1e3 / n_q, 4),
                         "note": "throughput with several queries in flight (one handle + stream per lane sharing the database image, create_lane); "
                                 "each query's own latency is `value` or longer"}
            if args.batched_sweep:
                # the same lanes, but the B queries of a batch share ONE pass over the database (spiral_gpu_server_first_dim_batch):
                # run_pre per lane on its stream, the batched sweep, run_post per lane
                pipelined["batched_sweep"] = {}
                for B in (2, 4):
                    group = [ln for ln, _ in lanes[:B]]
                    for _ in range(2):  # graph capture of run_pre / run_post, untimed
                        for ln in group: ln.run_pre()
                        sa.first_dim_batch(group)
                        for ln in group: ln.run_post()
                    torch.cuda.synchronize()
                    n_b = max(n_q // B, 3)
                    t1 = time.perf_counter()
                    for _ in range(n_b):
                        for ln in group: ln.run_pre()
                        sa.first_dim_batch(group)
                        for ln in group: ln.run_post()
                    torch.cuda.synchronize()
                    dt_b = time.perf_counter() - t1
                    pipelined["batched_sweep"][str(B)] = {"queries": n_b * B, "queries_per_s": round(n_b * B / dt_b, 1), "ms_per_query_amortised": round(dt_b * 1e3 / (n_b * B), 4)}
                pipelined["batched_sweep"]["note"] = ("B queries per pass over the database (sweep_mfma_kernel), each lane's expansion / conversion / fold still its own launches "
                                                      "on its own stream; throughput only, every answer bit-identical to the single-query path")
            if args.batched_sweep:
                # whole queries batched: every launch of the answer carries the B queries of a batch (spiral_gpu_server_run_query_batch: the
                # expansion / conversion / lift / fold launches take a query dimension, the sweep is the batched one), one hipGraph replay per batch
                bq = {}
                srv.set_acc(0)  # (back to the server's own accumulators: the lanes of a batch address their buffers relative to one another)
                for name, groups in (("2", [[0, 1]]), ("4", [[0, 1, 2, 3]]), ("8", [list(range(8))]), ("2x2", [[0, 1], [2, 3]]), ("2x8", [list(range(8)), list(range(8, 16))])):
                    for g in groups:  # the lanes of a batch on the batch's own stream (lane 0's): no cross-stream ordering around the launch sequence
                        for i in g: lanes[i][0].set_stream(lanes[g[0]][1].cuda_stream)
                    groups = [[lanes[i][0] for i in g] for g in groups]
                    for _ in range(2):  # graph capture, untimed
                        for g in groups: sa.run_query_batch(g)
                    torch.cuda.synchronize()
                    per = sum(len(g) for g in groups)
                    n_b = max(n_q // per, 3)
                    t1 = time.perf_counter()
                    for _ in range(n_b):
                        for g in groups: sa.run_query_batch(g)
                    torch.cuda.synchronize()
                    dt_b = time.perf_counter() - t1
                    bq[name] = {"queries": n_b * per, "queries_per_s": round(n_b * per / dt_b, 1), "ms_per_query_amortised": round(dt_b * 1e3 / (n_b * per), 4),
                                "ms_per_batch": round(dt_b * 1e3 / n_b / len(groups), 4)}
                bq["note"] = ("B whole queries (different lanes: own keys, own query) per launch sequence, every launch carrying all B (gridDim.z = B) and the sweep ONE pass "
                              "over the database for all of them on the matrix cores (sweep_mfma_kernel: i8 limb products, bit-identical accumulators); '2x2' / '2x8' = two such "
                              "batches of 2 / 8 in flight on two streams (one batch's HBM-bound sweep under the other's VALU-bound stages); throughput only -- a query's latency "
                              "is ms_per_batch (twice that with two batches in flight)")
                pipelined["batched_query"] = bq
                bqr = batch_roofline_from_profile(nu1, nu2)
                if bqr: pipelined["batched_query_roofline"] = bqr
                # the batched sweep's own roofline: one launch of sweep_mfma_kernel for B queries (HIP events on the launch stream, 12 launches), against
                # the bytes ONE pass has to move: the database once (SURVEY 8d's 8 bytes per word), B queries' records and accumulators
                sk = {}
                for ln, _ in lanes: ln.run_pre()
                torch.cuda.synchronize()
                for B in (2, 4, 5, 8):
                    group = [ln for ln, _ in lanes[:B]]
                    sa.time_sweep_batch(group, 2)
                    ms_b = sa.time_sweep_batch(group, 12)
                    one, one_dev = srv.sweep_bytes(), srv.sweep_device_bytes()
                    db_b = shp.dim0 // world * shp.num_per * 4 * 2048 * 8
                    alg, dev_b = db_b + B * (one - db_b), db_b // 8 * 7 + B * (one_dev - db_b // 8 * 7)
                    sk[str(B)] = {"avg_launch_ms": round(ms_b, 4), "algorithmic_bytes_per_launch": alg, "achieved": round(alg / (ms_b * 1e-3) / 1e9, 1),
                                  "frac": round(alg / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4), "device_bytes_per_launch": dev_b,
                                  "frac_device_bytes": round(dev_b / (ms_b * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                  "queries_x_database_bytes_per_s": round(B * db_b / (ms_b * 1e-3) / 1e9, 1)}
                sk["note"] = ("bound: hbm, peak %d GB/s; `achieved` / `frac` in algorithmic bytes (SURVEY 8d's 8 bytes per database word, the database counted ONCE per launch: the "
                              "image holds a word in 7 bytes, so `frac` can pass 1 where `frac_device_bytes` -- bytes physically moved -- is 0.88); per query the pass streams "
                              "`queries_x_database_bytes_per_s` GB/s of database" % HBM_PEAK_GBPS)
                pipelined["batched_sweep_kernel"] = sk
                # ONE database image: the batches above made the holder convert its image to limb planes IN PLACE (include/spiral_gpu.h,
                # spiral_gpu_server_set_db_format); single queries on such a server sweep it with the one-query instance of the matrix-core kernel
                image = {"format_after_batches": "limb planes" if srv.db_format() == SV.DB_LIMBS else "packed", "db_device_bytes": int(srv.db_device_bytes()),
                         "one_image_bytes": int(srv.sweep_device_bytes() - shp.dim0 // world * 48 * 2048 - shp.num_per * 6 * 2048 * 8)}
                if srv.db_format() == SV.DB_LIMBS:
                    srv.time_sweep(2)
                    ms_l = srv.time_sweep(12)
                    image["one_query_sweep_on_limb_planes"] = {"kernel": "sweep_mfma_kernel<1>", "avg_launch_ms": round(ms_l, 4), "frac": round(srv.sweep_bytes() / (ms_l * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4)}
                    for _ in range(3): srv.run_query()  # (re-captured for the new form)
                    torch.cuda.synchronize()
                    t1 = time.perf_counter()
                    for _ in range(20): srv.run_query()
                    torch.cuda.synchronize()
                    image["one_query_sweep_on_limb_planes"]["ms_per_query"] = round((time.perf_counter() - t1) * 1e3 / 20, 4)
                    image["one_query_sweep_on_limb_planes"]["answer_sha256"] = read_answer()
                image["note"] = ("db_device_bytes = device memory held for database images after the batched legs: one image (3.5 bytes per residue), not two; the image goes back to the "
                                 "packed form (in place) before the standalone kernel timings below")
                pipelined["database_image"] = image
            for lane, _ in lanes[1:]:
                lane.close()
            if srv.db_format() != SV.DB_PACKED:
                srv.set_db_format(SV.DB_PACKED)
        # untimed: the reference's stage buckets (src/spiral.cpp:246-257) from one eager pass with HIP events
        detail = None
        if primary and world == 1:
            srv.set_overlap(0)
            srv.use_graphs(False)
            detail = srv.answer_resident()
            detail = srv.answer_resident()

    prog.arm(f"{workload}/standalone kernels", 3.0)
    extra = {"reference_buckets_us_eager": ({k: round(v, 1) for k, v in detail.items() if k != "scaltomat_us"} if detail else None)}
    # the same kernel timed over 24 back-to-back launches on the server stream, outside the timed region (spiral_gpu_server_time_sweep): the
    # in-loop figure averages only the sampled steps (steps / event_every launches) and moves +-4 % with them; both are printed
    sweep_ms_24 = srv.time_sweep(24) if world == 1 else None
    if sweep_ms_24:
        bytes_sweep = srv.sweep_bytes()
        extra["roofline"] = {"standalone_24_launches": {"avg_launch_ms": round(sweep_ms_24, 4), "achieved": round(bytes_sweep / (sweep_ms_24 * 1e-3) / 1e9, 1),
                                                        "frac": round(bytes_sweep / (sweep_ms_24 * 1e-3) / 1e9 / HBM_PEAK_GBPS, 4),
                                                        "note": "24 back-to-back launches of the same kernel on the same inputs after the timed region (HIP events on the launch stream)"}}
    # the transform kernels against their VALU bound (they are the rest of the query: ~29 k limb-pair transforms at config 2).
    # Bound: the bare Harvey / Shoup butterfly = 3 integer multiplies (4.45 cycles each per wave instruction per SIMD, measured,
    # profiles/r02_ubench_valu.txt) + 4 add / sub (2.9) = 25 cycles x 88 limb-butterflies per thread, 4 waves per polynomial on
    # 4 SIMDs: 2200 cycles per polynomial per CU = 3.6 ns per polynomial over 256 CUs at 2.4 GHz.
    if primary and rank == 0 and world == 1:
        fwd_ms, inv_ms = sa.time_ntt(16384, 10)
        ns_f, ns_i = fwd_ms * 1e6 / 16384, inv_ms * 1e6 / 16384
        ns_d = sa.time_ntt_digits(2048, 8, 10) * 1e6 / 16384
        # Two bounds per launch, the larger one applies: VALU issue (3.6 ns, above) and HBM -- to_ntt / from_ntt read AND write 16 KiB per transform
        # (32 KiB / 8 TB/s = 4.1 ns), the digit launch writes 16 KiB and reads its source once per 8 digits (18 KiB / 8 TB/s = 2.3 ns: VALU-bound)
        hbm_rw, hbm_dig = 32768 / HBM_PEAK_GBPS, (16384 + 16384 / 8) / HBM_PEAK_GBPS  # ns per transform
        extra["roofline_ntt"] = {"bound": "max(valu, hbm) per launch", "unit": "ns per limb-pair transform (2048 points x 2 primes)", "peak": 3.6, "peak_valu": 3.6,
                                 "peak_hbm": {"forward_to_ntt": round(hbm_rw, 2), "inverse_from_ntt": round(hbm_rw, 2), "forward_digits": round(hbm_dig, 2)},
                                 "forward_to_ntt": round(ns_f, 2), "inverse_from_ntt": round(ns_i, 2), "forward_digits": round(ns_d, 2),
                                 "frac_forward": round(max(3.6, hbm_rw) / ns_f, 3), "frac_inverse": round(max(3.6, hbm_rw) / ns_i, 3), "frac_forward_digits": round(max(3.6, hbm_dig) / ns_d, 3),
                                 "frac_vs_valu_only": {"forward_to_ntt": round(3.6 / ns_f, 3), "inverse_from_ntt": round(3.6 / ns_i, 3), "forward_digits": round(3.6 / ns_d, 3)},
                                 "batch": 16384, "note": "standalone launches of 16384 transforms, HIP events.  frac_* = the LARGER of the two bounds / measured: to_ntt (reduce mod p, b + forward) "
                                                         "and from_ntt (inverse + CRT lift) each read and write 16 KiB of HBM per transform, so their bound is the memory one (4.1 ns at the "
                                                         "8 TB/s peak); forward_digits is the launch the stages are built from (8 gadget digits of each of 2048 polynomials: the source is "
                                                         "re-read from cache, every transform writes its 16 KiB) and is bound by VALU issue (3.6 ns)"}
    if pipelined: extra["pipelined"] = pipelined
    out = make_out(extra)
    prog.update(out)
    prog.disarm()
    srv.close()
    del acc
    torch.cuda.empty_cache()
    return out, params_kw


def bench_replicas(args, ctx, prog, batch=8, n_rounds=10, in_flight=2):
    """N > 1, beside the j-shard figure north_star prescribes (never `value`, never the scaling curve): every rank answers on its OWN full copy of the
    configs[1] database, `batch` whole queries per launch sequence (spiral_gpu_server_run_query_batch), `in_flight` such batches on their own streams (one
    batch's HBM-bound sweep runs under the other's VALU-bound stages) -- what a deployment would do at this database size, where the j-shard is Amdahl-bound
    (DESIGN.md section 6).  No collective on the data path: a barrier either side, max over ranks."""
    import numpy as np

    import spiral_amd as sa

    torch = ctx.torch
    params_kw = {k: v for k, v in WORKLOADS["config2"].items() if k != "label"}
    pg = sa.make_params(**params_kw)
    shp = sa.get_shape(pg)
    prog.arm("replicas/set-up", 3.0)
    owner = sa.Server(pg, ctx.local_rank)
    streams = [torch.cuda.Stream(device=ctx.dev) for _ in range(in_flight)]
    owner.set_stream(streams[0].cuda_stream)
    owner.gen_db(DB_SEED)
    lanes = [owner] + [sa.Server(pg, ctx.local_rank, share_db_of=owner) for _ in range(in_flight * batch - 1)]
    groups = [lanes[g * batch:(g + 1) * batch] for g in range(in_flight)]
    pub, query = synth_inputs(np, sa, pg, shp)
    for g, grp in enumerate(groups):
        for ln in grp:
            ln.set_stream(streams[g].cuda_stream)  # the lanes of a batch on the batch's stream: no cross-stream ordering around the launch sequence
            ln.set_pub_params(*pub)
            ln.set_query(query)
            ln.use_graphs(True)
    prog.arm("replicas/timed")
    for _ in range(4):
        for grp in groups: sa.run_query_batch(grp)
    ctx.fence()
    t0 = time.perf_counter()
    for _ in range(n_rounds):
        for grp in groups: sa.run_query_batch(grp)
    ctx.fence()
    dt = ctx.max_over_ranks(time.perf_counter() - t0)
    for ln in lanes[1:]:
        ln.close()
    owner.close()
    torch.cuda.empty_cache()
    per_rank = in_flight * batch * n_rounds
    return {"n_replicas": ctx.world, "batch": batch, "batches_in_flight": in_flight, "queries": ctx.world * per_rank, "queries_per_s": round(ctx.world * per_rank / dt, 1),
            "queries_per_s_per_replica": round(per_rank / dt, 1), "ms_per_batch": round(dt * 1e3 / n_rounds, 4),
            "note": "N independent replicas of the configs[1] database, one per GPU, each answering batches of whole queries (two batches in flight on two streams; "
                    "ms_per_batch = one round of both); throughput only, NOT the j-shard scaling north_star asks for (that is `value` at each N) and not a latency"}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC, which RCCL needs on this driver: set before torch is imported, also under an external launcher
    args = parse_args(argv)
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        raise SystemExit(self_launch(args, argv))
    if args.workload == "pack":
        return bench_pack(args)

    import numpy as np

    prog = Progress(int(os.environ.get("RANK", "0")), args.watchdog)
    prog.catch_sigterm()
    try:
        return run(args, prog, np)
    except BaseException as e:  # a rank that fails takes the job down (the launcher SIGTERMs the others): rank 0 prints what it has first
        if isinstance(e, SystemExit) and not e.code:
            raise
        import traceback

        traceback.print_exc()
        prog.give_up(1, error=repr(e), failed_in=prog.where)


def extra_leg(prog, name, fn):
    """a leg beside the headline (replicas, the secondary geometry, the whole-item stream, the CPU baseline): its failure is recorded in the line, it does not cost the line"""
    try:
        return fn(), None
    except Exception as e:
        import traceback

        traceback.print_exc()
        print(f"bench.py: leg '{name}' failed on rank {prog.rank}: {e!r}; the line goes on without it", file=sys.stderr, flush=True)
        try:
            import torch

            torch.cuda.empty_cache()
        except Exception:
            pass
        return None, {"error": repr(e)[:400], "failed_in": prog.where}


def run(args, prog, np):
    prog.arm("process group set-up", 3.0)
    ctx = Ctx(args, prog)
    prog.arm("set-up", 3.0)
    if args.workload == "stream":
        out = bench_stream(args, ctx, prog)
        prog.arm("process group tear-down", 1.0)
        ctx.close()
        prog.disarm()
        if ctx.rank == 0:
            import ctypes

            ctypes.CDLL(None).fflush(None)
            prog.done = True
            print(json.dumps(out), flush=True)
        return
    if args.workload == "stream-instance":
        args.workload = "stream"
    out, params_kw = bench_base(args, ctx, args.workload, args.steps, args.warmup, True, prog)
    if ctx.world > 1 and "config2" in (args.workload, args.secondary) and not args.no_replicas and (args.nu1, args.nu2) == (None, None):
        rep, err = extra_leg(prog, "replicas", lambda: bench_replicas(args, ctx, prog))  # no collective on its data path: cannot hang where the j-shard schedules did not
        out["replicas"] = rep if err is None else err
        prog.update(out)
    if args.secondary and not args.no_config3 and (args.nu1, args.nu2) == (None, None):
        # secondary leg, LAST.  Default: configs[2]'s geometry (where the sweep is ~75 % of the query and the j-shard scales) under `also.config3`, the
        # headline configs[1] complete and in the line-so-far before it starts; --headline config3 swaps the two
        sec = args.secondary

        def nest(o3):
            c3 = {k: o3[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "stages_us", "queries_per_s", "schedules", "collectives_us", "pipelined", "answer_sha256", "value_no_prewarm") if k in o3}
            c3["workload"] = o3["config"]["workload"]
            c3["parallelism"] = o3["config"]["parallelism"]
            if "roofline" in o3:
                c3["roofline"] = {k: o3["roofline"][k] for k in ("achieved", "frac", "frac_device_bytes", "avg_launch_ms", "algorithmic_bytes_per_launch", "shard")}
            line = dict(out, also={sec: c3})
            # (the driver keeps `config` whole: the secondary leg's value and answer hash are readable from its record too)
            line["config"] = dict(out["config"], **{f"also_{sec}_value": c3.get("value"), f"also_{sec}_answer_sha256": c3.get("answer_sha256")})
            return line

        prog.wrap = nest
        o3, err = extra_leg(prog, "also." + sec, lambda: bench_base(args, ctx, sec, args.config3_steps if sec == "config3" else args.steps, min(args.warmup, 2) if sec == "config3" else args.warmup, False, prog)[0])
        prog.wrap = lambda line: line
        out = nest(o3) if err is None else dict(out, also={sec: err})
    if args.workload in ("config2", "config3") and not args.no_stream_item and not args.no_config3 and (args.nu1, args.nu2) == (None, None):  # (--no-config3 = no extra legs at all)
        # last leg: configs[3] WHOLE (bench_stream): one query against the 7 instances of a 100 KB item's database; rank r holds instances r, r + N, ...
        prog.update(out)
        st, err = extra_leg(prog, "also.stream_item", lambda: bench_stream(args, ctx, prog, steps=3, warmup=1))
        if err is not None:
            out = dict(out, also=dict(out.get("also", {}), stream_item=err))
        elif ctx.rank == 0:
            out = dict(out, also=dict(out.get("also", {}), stream_item={k: st[k] for k in ("value", "unit", "n_gpus", "steps", "warmup", "item", "answer_sha256")}))
            out["also"]["stream_item"]["workload"] = st["config"]["workload"]
            out["also"]["stream_item"]["parallelism"] = st["config"]["parallelism"]
            out["also"]["stream_item"]["roofline"] = {k: st["roofline"][k] for k in ("kernel", "achieved", "frac", "frac_device_bytes", "avg_launch_ms", "database_bytes_per_s_over_the_item_query")}
            out["config"] = dict(out["config"], also_stream_item_value=st["value"], also_stream_item_answer_sha256=st["answer_sha256"])
    prog.update(out)
    prog.arm("process group tear-down", 1.0)
    ctx.close()
    prog.disarm()
    if ctx.rank == 0:
        if ctx.world == 1 and not args.no_cpu_baseline and args.workload == "config2" and args.headline != "config3":  # (the larger workloads' 32 / 64 GiB host databases are not built for a baseline)
            cb, err = extra_leg(prog, "cpu_baseline", lambda: cpu_baseline(params_kw, np, args.workload))
            out["cpu_baseline"] = cb if err is None else err
        import ctypes

        ctypes.CDLL(None).fflush(None)  # RCCL printf()s a banner into C stdio; get it out before the JSON
        sys.stdout.flush()
        prog.done = True
        print(json.dumps(out), flush=True)  # the ONE JSON line, last thing on stdout


if __name__ == "__main__":
    main()

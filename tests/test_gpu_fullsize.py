"""Bit-exact parity at BASELINE.json's full sizes, through the C ABI (`-m gpu`, marked slow).

The oracle side runs on the `oracle_mt` fixture: the same oracle sources built on the test machine with OpenMP
(tests/test_oracle.py proves that build equal to the default one), so that a 2 GiB database and a full-size query are
seconds of host time.  What is compared bit for bit:

* configs[1] / configs[0] (nu1=8, nu2=7, 2^20 x 256 B): every stage output -- expanded ciphertexts, ScalToMat outputs, GSW
  matrices, the 24 MiB of sweep accumulators, the lifted and folded ciphertexts, the 96 KiB response -- from the stage
  API, from answer() and from the whole-query hipGraph; with the database generated on the device, uploaded in the
  reference's NTT layout, and ingested from raw plaintext bytes (SURVEY.md 8c asked for the SHA-256 of the sweep
  output and of the response: printed, and equal because the arrays are).
* configs[2] geometry (nu1=9, nu2=10, 32 GiB) and configs[3] (SpiralStream nu1=11, nu2=9, 64 GiB): the database does not
  fit a host-side reference computation, and the sweep is independent per NTT slot and per output ciphertext, so the device
  database is read back (a) for a sample of slots -- the oracle runs the sweep on exactly those slots for ALL output
  ciphertexts -- and (b) for ALL 2048 slots of the first, the last and one random plaintext column -- the oracle runs the
  full multiplyQueryByDatabase on them, every z-tile and XCD of the sweep's tile map included -- and the accumulators must
  agree; sampled items of the device database must equal the oracle's encoding of the seeded plaintexts; everything
  database-independent (expansion, conversion) is compared in full; the response must decode to the item.
* configs[4] (SpiralPack nu1=10, nu2=8, n=4, 64 GiB): the full sweep accumulators of trials 0, 7 and 15 against the oracle's
  sweep of each trial's 4 GiB database, unconditionally; with 160 GiB of host memory, all 16 trials at once, the packed
  ciphertext and the response.
The lines the tests record (SHA-256s, GB/s, which branches ran) are repeated in pytest's terminal summary (tests/conftest.py).
"""
import hashlib

import numpy as np
import pytest

pytestmark = [pytest.mark.gpu, pytest.mark.slow]
N = 2048


@pytest.fixture(scope="module")
def sa():
    import torch  # torch first: it ships its own HIP runtime

    torch.cuda.is_available()
    import spiral_amd

    assert spiral_amd.lib().spiral_gpu_device_count() > 0, "GPU tests need a device"
    return spiral_amd


def assert_eq(got, exp, what):
    if not (got.shape == exp.shape and (got == exp).all()):
        bad = np.argwhere(got != exp) if got.shape == exp.shape else []
        raise AssertionError(f"{what}: shapes {got.shape} / {exp.shape}, {len(bad)} of {got.size} words differ, first at {bad[:5].tolist() if len(bad) else '-'}")


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a, dtype="<u8").tobytes()).hexdigest()[:16]


def host_gib_available():
    for line in open("/proc/meminfo"):
        if line.startswith("MemAvailable:"):
            return int(line.split()[1]) / 2**20
    return 0.0


def record(request, line):
    """print, and keep for the terminal summary (tests/conftest.py) so that the line survives `pytest -q`"""
    print(line)
    request.config._spiral_evidence.append(line)


def test_config2_every_stage_bit_exact(sa, oracle_mt, request):
    """configs[1] (= configs[0]'s geometry): the whole path at 2^20 x 256 B against the oracle, stage by stage"""
    M = oracle_mt
    from spiral_amd import server as SV

    po, pg = M.make_params(8, 7), sa.make_params(8, 7)
    s = M.shape_of(po)
    cl = M.Client(po, seed=1)
    wl, wr, w, v = cl.pub_params()
    idx = 1234
    q = cl.query(idx)
    db = M.gen_db(po, 1234)  # 2 GiB, the reference's layout
    cv = M.stage_expand(po, q, wl, wr)
    cts, gsw = M.stage_convert(po, cv, w, v)
    acc = M.multiply_query_by_database(M.reorient_ciphertexts(cts), db, s.dim0, s.num_per)
    raw = M.from_ntt(acc)
    fin = M.stage_fold(po, raw, gsw)
    resp = M.stage_rescale(po, fin)
    record(request, f"config 2 oracle ({M.n_threads} threads): sha256 sweep output {sha(acc)} response {sha(resp)}")

    srv = sa.Server(pg)
    srv.keep_cts(True)
    srv.gen_db(1234)
    srv.set_pub_params(wl, wr, w, v)
    srv.set_query(q)
    srv.expand()
    assert_eq(srv.read(SV.BUF_EXPANDED), cv, "expanded ciphertexts")
    srv.convert()
    assert_eq(srv.read(SV.BUF_CTS), cts, "scalToMat outputs")
    assert_eq(srv.read(SV.BUF_GSW), gsw, "regevToGSW outputs")
    srv.first_dim()
    got_acc = srv.read(SV.BUF_ACC)
    assert_eq(got_acc, acc, "first-dimension accumulators (24 MiB)")
    srv.lift()
    assert_eq(srv.read(SV.BUF_RAW), raw, "lifted first-dimension result")
    srv.fold()
    assert_eq(srv.read(SV.BUF_FINAL), fin, "folded ciphertext")
    srv.finish()
    got_resp = srv.read(SV.BUF_RESPONSE)
    assert_eq(got_resp, resp, "response")
    record(request, f"config 2 device: sha256 sweep output {sha(got_acc)} response {sha(got_resp)} (all 24 MiB of accumulators, every stage buffer, answer(), run_query graph: bit-exact)")
    assert_eq(cl.decode(got_resp), M.db_item(po, 1234, idx), "decoded plaintext")

    # answer() in one call, and the whole query replayed as one hipGraph, for other indices too
    for i2 in (0, (1 << 15) - 1):
        q2 = cl.query(i2)
        want = M.answer(po, q2, wl, wr, w, v, db)
        f2, r2, _ = srv.answer(q2)
        assert_eq(f2, want, f"answer() idx={i2}")
        assert_eq(r2, M.stage_rescale(po, want), f"answer() response idx={i2}")
    srv.keep_cts(False)
    srv.use_graphs(True)
    for rep in range(2):
        srv.set_query(q)
        srv.run_query()
        srv.sync()
        assert_eq(srv.read(SV.BUF_ACC), acc, "run_query graph: accumulators")
        assert_eq(srv.read(SV.BUF_FINAL), fin, "run_query graph: folded ciphertext")
        assert_eq(srv.read(SV.BUF_RESPONSE), resp, "run_query graph: response")
    srv.use_graphs(False)

    # bench.py's own synthetic inputs (uniform residues from default_rng(1), database seed 1234): the oracle's answer for exactly what the timed
    # hipGraph computes, as the hash bench.py prints -- BENCH_rNN.json's answer_sha256 must equal this line
    import bench

    pub_b, q_b = bench.synth_inputs(np, sa, pg, s)
    assert bench.DB_SEED == 1234 and bench.WORKLOADS["config2"]["nu1"] == 8 and bench.WORKLOADS["config2"]["nu2"] == 7
    assert all(getattr(sa.make_params(**{k: v for k, v in bench.WORKLOADS["config2"].items() if k != "label"}), f) == getattr(pg, f)
               for f in ("nu1", "nu2", "t_gsw", "t_conv", "t_exp", "t_exp_right", "qprime_bits", "p_db", "direct_upload"))
    fin_b = M.answer(po, q_b, *pub_b, db)
    resp_b = M.stage_rescale(po, fin_b)
    srv.use_graphs(True)
    srv.set_pub_params(*pub_b)
    srv.set_query(q_b)
    for rep in range(3):
        srv.run_query()
    srv.sync()
    assert_eq(srv.read(SV.BUF_FINAL), fin_b, "bench.py's inputs: folded ciphertext of the replayed whole-query graph")
    assert_eq(srv.read(SV.BUF_RESPONSE), resp_b, "bench.py's inputs: response")
    record(request, f"bench.py config2 inputs (default_rng(1) residues, gen_db(1234)): oracle answer_sha256 {bench.answer_hash(np, fin_b, resp_b)} == device "
                    f"{bench.answer_hash(np, srv.read(SV.BUF_FINAL), srv.read(SV.BUF_RESPONSE))}")
    srv.use_graphs(False)
    srv.set_pub_params(wl, wr, w, v)

    # eight, four and two whole queries in one launch sequence (run_query_batch, every launch carrying all of them; seven more clients with their own keys on lanes
    # of the same database image): every lane's accumulators, folded ciphertext and response against the oracle's for ITS inputs
    lanes = [srv] + [sa.Server(pg, share_db_of=srv) for _ in range(7)]
    clients = [cl] + [M.Client(po, seed=100 + b) for b in range(7)]
    pps = [(wl, wr, w, v)] + [c.pub_params() for c in clients[1:]]
    idxs = [idx, 0, (1 << 15) - 1, 7777, 1, 12345, 31000, 2048]
    qs_b = [c.query(i) for c, i in zip(clients, idxs)]
    for ln, pp, qq in zip(lanes, pps, qs_b):
        ln.set_pub_params(*pp)
        ln.set_query(qq)
        ln.use_graphs(True)
    for n_b in (8, 4, 2):  # (every batch one pass over the database on the matrix cores: 6, 3 and 2 column tiles)
        for rep in range(2):  # capture, then a replay
            sa.run_query_batch(lanes[:n_b])
        for b, (ln, c, pp, qq, i2) in enumerate(zip(lanes[:n_b], clients, pps, qs_b, idxs)):
            ln.sync()
            cv_b = M.stage_expand(po, qq, pp[0], pp[1])
            cts_b, gsw_b = M.stage_convert(po, cv_b, pp[2], pp[3])
            acc_b = M.multiply_query_by_database(M.reorient_ciphertexts(cts_b), db, s.dim0, s.num_per)
            fin_l = M.stage_fold(po, M.from_ntt(acc_b), gsw_b)
            assert_eq(ln.read(SV.BUF_ACC), acc_b, f"batch of {n_b}, lane {b}: accumulators")
            assert_eq(ln.read(SV.BUF_FINAL), fin_l, f"batch of {n_b}, lane {b}: folded ciphertext")
            got_r = ln.read(SV.BUF_RESPONSE)
            assert_eq(got_r, M.stage_rescale(po, fin_l), f"batch of {n_b}, lane {b}: response")
            assert_eq(c.decode(got_r), M.db_item(po, 1234, i2), f"batch of {n_b}, lane {b}: decoded plaintext")
    record(request, "config 2 run_query_batch: 8, 4 and 2 queries per launch sequence (own keys per lane), every lane's accumulators / folded ciphertext / response bit-exact")
    for ln in lanes[1:]:
        ln.close()
    srv.use_graphs(False)

    # the same database uploaded in the reference's NTT layout (load_db), then ingested from raw plaintext bytes
    assert_eq(srv.read_db_slots(5, 2).reshape(2, -1), db.reshape(N, -1)[5:7], "device-generated database, slots 5..6")
    srv.load_db(db)
    srv.set_query(q)
    srv.expand()
    srv.convert()
    srv.first_dim()
    assert_eq(srv.read(SV.BUF_ACC), acc, "accumulators from the uploaded database")
    pts = np.stack([M.db_item(po, 1234, i) for i in range(s.dim0 * s.num_per)])  # 256 MiB of plaintext bytes at p = 256
    srv.fill_db_random(1)  # make sure the ingest really rebuilds it
    srv.load_db_items(M.pack_items(pts, 8), 8)
    srv.first_dim()
    assert_eq(srv.read(SV.BUF_ACC), acc, "accumulators from the database ingested from plaintext bytes")
    srv.close()


def sampled_checks(sa, M, po, pg, seed, idx, cl, label, request, n_slots=6, n_items=12, batches=()):
    """the size-independent checks for a database too large for a host-side reference: see the module docstring"""
    from spiral_amd import server as SV

    s = M.shape_of(po)
    total = s.dim0 * s.num_per
    wl, wr, w, v = cl.pub_params()
    q = cl.query(idx)
    srv = sa.Server(pg)
    srv.keep_cts(True)
    srv.gen_db(seed)
    srv.set_pub_params(wl, wr, w, v)
    fin, resp, us = srv.answer(q)
    assert_eq(cl.decode(resp), M.db_item(po, seed, idx), f"{label}: decoded plaintext")
    # database-independent stages in full
    cv = M.stage_expand(po, q, wl, wr)
    cts, gsw = M.stage_convert(po, cv, w, v)
    assert_eq(srv.read(SV.BUF_EXPANDED), cv, f"{label}: expanded ciphertexts")
    assert_eq(srv.read(SV.BUF_CTS), cts, f"{label}: scalToMat outputs")
    assert_eq(srv.read(SV.BUF_GSW), gsw, f"{label}: regevToGSW outputs")
    # sampled items of the device database == the oracle's encoding of the seeded plaintexts
    rng = np.random.default_rng(seed)
    for item in [0, total - 1, idx] + [int(x) for x in rng.integers(0, total, size=n_items)]:
        assert_eq(srv.read_db_item(item), M.encode_item(po, M.db_item(po, seed, item)), f"{label}: database item {item}")
    # the sweep on sampled slots: the oracle evaluates exactly those slots on the device's own database slabs
    got_acc = srv.read(SV.BUF_ACC)
    re = M.reorient_ciphertexts(cts)
    zs = sorted({0, N - 1} | {int(x) for x in rng.integers(0, N, size=n_slots)})
    slabs = np.concatenate([srv.read_db_slots(z, 1) for z in zs]).reshape(len(zs), -1)
    want = M.multiply_query_by_database_slots(re[zs], slabs, s.dim0, s.num_per)
    assert_eq(got_acc[..., zs], want, f"{label}: first-dimension accumulators on slots {zs}")
    # slot-complete: the FULL output polynomials (all 2048 slots, so every z-tile of the sweep and every XCD the tile map
    # deals them to) of the first, the last and one random plaintext column -- the device's database words of those columns
    # (dim0 x 2 x 2 x 2048 each) read back, multiplyQueryByDatabase (src/spiral.cpp:628-999) restated on them
    cols = sorted({0, s.num_per - 1, int(rng.integers(1, max(2, s.num_per - 1)))})
    for ii in cols:
        sub = srv.read_db_columns(ii, 1)  # load_db's layout with num_per = 1
        want_ii = M.multiply_query_by_database(re, sub, s.dim0, 1)
        assert_eq(got_acc[ii:ii + 1], want_ii, f"{label}: all 2048 slots of output ciphertext {ii}")
    record(request, f"{label}: sweep slot-complete on output ciphertexts {cols} (all 2048 slots x 6 polynomials each), on slots {zs} for all {s.num_per} ciphertexts; sha256 of those columns {sha(got_acc[cols])}")
    # the batched sweep (sweep_kernel<0, B>: B queries per pass over the database) at this geometry: every lane's accumulators against
    # the same sampled slots and slot-complete columns, each lane with its own query
    if batches:
        import torch

        nb = max(batches)
        lanes, res, gsws, qbs = [srv], [re], [gsw], [q]
        for b in range(1, nb):
            lane = sa.Server(pg, share_db_of=srv)
            lane.set_pub_params(wl, wr, w, v)
            qb = cl.query((idx * (b + 1) + 12345 * b) % total)
            lane.set_query(qb)
            lanes.append(lane)
            cts_b, gsw_b = M.stage_convert(po, M.stage_expand(po, qb, wl, wr), w, v)
            res.append(M.reorient_ciphertexts(cts_b))
            gsws.append(gsw_b)
            qbs.append(qb)
        streams = [torch.cuda.Stream() for _ in lanes]
        for ln, st in zip(lanes, streams):
            ln.set_stream(st.cuda_stream)
        srv.set_query(q)
        subs = {ii: srv.read_db_columns(ii, 1) for ii in cols}
        for B in batches:
            for ln in lanes[:B]:
                ln.run_pre()
            sa.first_dim_batch(lanes[:B])
            for b, ln in enumerate(lanes[:B]):
                ln.sync()
                acc_b = ln.read(SV.BUF_ACC)
                assert_eq(acc_b[..., zs], M.multiply_query_by_database_slots(res[b][zs], slabs, s.dim0, s.num_per), f"{label}: batch of {B}, lane {b}: accumulators on slots {zs}")
                for ii in cols:
                    assert_eq(acc_b[ii:ii + 1], M.multiply_query_by_database(res[b], subs[ii], s.dim0, 1), f"{label}: batch of {B}, lane {b}: all slots of output ciphertext {ii}")
                if b == 0:
                    assert_eq(acc_b, got_acc, f"{label}: batch of {B}, lane 0 == its single sweep")
        record(request, f"{label}: batched sweeps B = {list(batches)}: every lane's accumulators bit-exact on slots {zs} (all ciphertexts) and slot-complete on ciphertexts {cols}")
        # the whole answer for all lanes in one launch sequence (run_query_batch) at this geometry: the accumulators must be those of the batched sweep
        # just checked, and each lane's folded ciphertext / response the oracle's fold of ITS accumulators with ITS GSW matrices
        accs = [ln.read(SV.BUF_ACC).copy() for ln in lanes]
        srv.keep_cts(False)
        for ln, qb in zip(lanes, qbs):
            ln.set_query(qb)
            ln.use_graphs(True)
        for rep in range(2):  # capture, then a replay
            sa.run_query_batch(lanes)
        for b, ln in enumerate(lanes):
            ln.sync()
            assert_eq(ln.read(SV.BUF_ACC), accs[b], f"{label}: whole-query batch of {nb}, lane {b}: accumulators == the batched sweep's")
            want_b = M.stage_fold(po, M.from_ntt(accs[b]), gsws[b])
            assert_eq(ln.read(SV.BUF_FINAL), want_b, f"{label}: whole-query batch of {nb}, lane {b}: folded ciphertext")
            assert_eq(ln.read(SV.BUF_RESPONSE), M.stage_rescale(po, want_b), f"{label}: whole-query batch of {nb}, lane {b}: response")
            ln.use_graphs(False)
        record(request, f"{label}: run_query_batch of {nb} lanes: every lane's accumulators, folded ciphertext and response bit-exact")
        srv.keep_cts(True)
        for ln in lanes[1:]:
            ln.close()
        srv.set_stream(0)
    # and everything after the sweep from the device's full accumulators
    raw = M.from_ntt(got_acc)
    want_fin = M.stage_fold(po, raw, gsw)
    assert_eq(fin, want_fin, f"{label}: folded ciphertext (from the device's accumulators)")
    assert_eq(resp, M.stage_rescale(po, want_fin), f"{label}: response")
    gbps = srv.sweep_bytes() / us["sweep_kernel_us"] / 1e3
    record(request, f"{label}: stage us {({k: round(x) for k, x in us.items()})}, sweep {gbps:.0f} GB/s of algorithmic bytes = {gbps / 80:.1f} % of the 8 TB/s HBM peak")
    srv.close()


def bench_inputs_hash(sa, M, workload, label, request, n_slots=4):
    """bench.py's own synthetic inputs for `workload` (default_rng(1) residues, gen_db(1234)) at a size with no host-side reference database: the hash
    bench.py prints (`answer_sha256`, also inside `config`) next to the oracle's.  The oracle computes expansion and conversion in full, the sweep on sampled
    slots and on one complete output ciphertext from the device's own database words, and folding + response switch from the device's accumulators."""
    import bench
    from spiral_amd import server as SV

    kw = {k: v for k, v in bench.WORKLOADS[workload].items() if k != "label"}
    po, pg = M.make_params(**kw), sa.make_params(**kw)
    s = M.shape_of(po)
    pub, q = bench.synth_inputs(np, sa, pg, sa.get_shape(pg))
    srv = sa.Server(pg)
    srv.gen_db(bench.DB_SEED)
    srv.set_pub_params(*pub)
    srv.set_query(q)
    srv.use_graphs(True)
    for rep in range(2):
        srv.run_query()
    srv.sync()
    acc, fin, resp = srv.read(SV.BUF_ACC), srv.read(SV.BUF_FINAL), srv.read(SV.BUF_RESPONSE)
    cts, gsw = M.stage_convert(po, M.stage_expand(po, q, pub[0], pub[1]), pub[2], pub[3])
    re = M.reorient_ciphertexts(cts)
    rng = np.random.default_rng(5)
    zs = sorted({0, N - 1} | {int(x) for x in rng.integers(0, N, size=n_slots)})
    slabs = np.concatenate([srv.read_db_slots(z, 1) for z in zs]).reshape(len(zs), -1)
    assert_eq(acc[..., zs], M.multiply_query_by_database_slots(re[zs], slabs, s.dim0, s.num_per), f"{label}: bench.py's inputs: accumulators on slots {zs}")
    ii = int(rng.integers(0, s.num_per))
    assert_eq(acc[ii:ii + 1], M.multiply_query_by_database(re, srv.read_db_columns(ii, 1), s.dim0, 1), f"{label}: bench.py's inputs: all slots of output ciphertext {ii}")
    want_fin = M.stage_fold(po, M.from_ntt(acc), gsw)
    want_resp = M.stage_rescale(po, want_fin)
    assert_eq(fin, want_fin, f"{label}: bench.py's inputs: folded ciphertext of the replayed whole-query graph")
    assert_eq(resp, want_resp, f"{label}: bench.py's inputs: response")
    record(request, f"bench.py {workload} inputs (default_rng(1) residues, gen_db(1234)): oracle answer_sha256 {bench.answer_hash(np, want_fin, want_resp)} == device "
                    f"{bench.answer_hash(np, fin, resp)} (oracle: expansion, conversion, fold, switch in full; sweep on slots {zs} + ciphertext {ii} complete)")
    srv.close()


def test_config3_geometry_sampled_slots(sa, oracle_mt, request):
    """configs[2]'s geometry on one MI355X: 2^24 x 256 B, nu1=9, nu2=10, t_GSW=10, q'=2^22, 32 GiB"""
    M = oracle_mt
    kw = dict(t_gsw=10, t_conv=4, t_exp=8, t_exp_right=56, qprime_bits=22, p_db=256)
    po, pg = M.make_params(9, 10, **kw), sa.make_params(9, 10, **kw)
    sampled_checks(sa, M, po, pg, 99, 424242 % (1 << 19), M.Client(po, seed=6), "config 3 geometry (32 GiB)", request, batches=(2, 4))
    # BENCH_rNN.json's also.config3.answer_sha256 (= config.also_config3_answer_sha256) must equal this line
    bench_inputs_hash(sa, M, "config3", "config 3 geometry (32 GiB)", request)


def test_configs3_spiralstream_2_20_x_100KB(sa, oracle_mt, request):
    """configs[3]: SpiralStream (--direct-upload), 2^20 x 100 KB -- the published "Streaming 20/spiralstream" set
    (all_parameter_choices.txt:1149-1163: nu1=11, nu2=9, p=32768, q'=27 bits, t_GSW=4, t_conv=56, t_exp=2; direct upload =
    QNUMFIRST 2^nu1, QNUMREST t_GSW*nu2, src/spiral.cpp:2060-2061): 2048 + 36 uploaded ciphertexts, no expansion, one
    64 GiB instance of the database on the device, the sweep HBM-bound"""
    M = oracle_mt
    kw = dict(t_gsw=4, t_conv=56, t_exp=2, t_exp_right=56, qprime_bits=27, p_db=32768, direct_upload=1)
    po, pg = M.make_params(11, 9, **kw), sa.make_params(11, 9, **kw)
    s = M.shape_of(po)
    assert (s.n_query_cts, s.dim0, s.num_per) == (2048 + 36, 2048, 512)
    sampled_checks(sa, M, po, pg, 777, 31337 % (1 << 20), M.Client(po, seed=4), "configs[3] SpiralStream 2^20 x 100KB (64 GiB)", request, n_slots=4, n_items=8)


def test_configs3_item_of_three_instances_at_full_size(sa, oracle_mt, request):
    """configs[3] WHOLE at its real geometry: an item = `factor` instances of the 2^20-plaintext database (select_params.py:297-298; 7 for 100 KB items, three
    of them here: 3 x 56 GiB resident on the one device).  ONE query, converted once, answered against the three 64 GiB instances by one
    spiral_gpu_server_run_query_instances sequence (host-buffer form): every plaintext of the item decodes through the oracle's client, every response is the
    oracle's modulus switch of its folded ciphertext, and instance 0's folded ciphertext and response equal answer()'s on the same server (whose stages the test
    above checks against the oracle slot by slot at this size)."""
    M = oracle_mt
    from spiral_amd import server as SV

    kw = dict(t_gsw=4, t_conv=56, t_exp=2, t_exp_right=56, qprime_bits=27, p_db=32768, direct_upload=1)
    po, pg = M.make_params(11, 9, **kw), sa.make_params(11, 9, **kw)
    factor, idx = 3, 271828 % (1 << 20)
    cl = M.Client(po, seed=14)
    pp = cl.pub_params()
    q = cl.query(idx)
    inst = []
    for k in range(factor):
        sv = sa.Server(pg)
        sv.gen_db(4000 + k)
        inst.append(sv)
    inst[0].set_pub_params(*pp)
    inst[0].use_graphs(True)
    image = inst[0].db_device_bytes()
    for rnd in range(2):  # capture, replay
        resp, fin, us = inst[0].answer_instances(inst, q)
    for k in range(factor):
        assert_eq(cl.decode(resp[k]), M.db_item(po, 4000 + k, idx), f"plaintext {k} of item {idx} (instance seeded {4000 + k})")
        assert_eq(resp[k], M.stage_rescale(po, fin[k]), f"instance {k}: response = the switch of its folded ciphertext")
    inst[0].use_graphs(False)
    fin0, resp0, _ = inst[0].answer(q)
    assert_eq(fin[0], fin0, "instance 0: the item query's folded ciphertext == answer()'s")
    assert_eq(resp[0], resp0, "instance 0: response == answer()'s")
    assert sum(sv.db_device_bytes() for sv in inst) == factor * image
    record(request, f"configs[3] whole, {factor} of 7 instances at full size ({factor} x {image / 2**30:.0f} GiB on the device): one query, every plaintext of the item decoded; "
                    f"item query {us / 1e3:.1f} ms on the device = conversion + {factor} x (sweep + fold + switch); sha256 of the {factor} responses {sha(resp)}")
    for sv in inst:
        sv.close()


def test_config5_pack_bit_exact(sa, oracle_mt, request):
    """configs[4]: SpiralPack 2^18 x 30 KB (all_parameter_choices.txt:610-624: nu1=10, nu2=8, n=4, p=256, q'=2^20, t_GSW=8,
    t_conv=4, t_exp=16), 16 trial databases of 4 GiB"""
    M = oracle_mt
    kw = dict(t_gsw=8, t_conv=4, t_exp=16, t_exp_right=56, qprime_bits=20, p_db=256)
    po, pg = M.make_params(10, 8, **kw), sa.make_params(10, 8, **kw)
    out_n, seed = 4, 2024
    s = M.pack_shape_of(po, out_n)
    cl = M.PackClient(po, out_n, seed=12)
    wl, wr, v, vw = cl.pub_params()
    srv = sa.PackServer(pg, out_n)
    srv.gen_db(seed)
    srv.set_pub_params(wl, wr, v, vw)
    idx = 123456 % (1 << 18)
    q = cl.query(idx)
    resp, packed, us = srv.answer(q)
    assert_eq(cl.decode(resp), M.pack_db_item(po, out_n, seed, idx), "decoded items (config 5)")
    # trial 0: the sweep accumulators against the oracle's sweep of that trial's 4 GiB database
    cv = np.zeros((1 << s.g, 2, 2, N), dtype=np.uint64)
    cv[0] = q.reshape(2, 2, N)
    cv = M.expand_improved(cv, s.g, po.t_exp, wl, po.t_exp_right, wr, s.n_right, s.ell * po.nu2, s.stopround)
    re = M.reorient_dim1(cv, s.dim0, 2)
    # ALL trials, one 4 GiB database on the host at a time (no 64 GiB host image, whatever the box's memory): each trial's FULL sweep
    # output (all slots, all ciphertexts) against the oracle's sweep of that trial's database, then the oracle's own lift + fold of the
    # trial (foldCiphertextsDim1, src/testing.cpp:596-624); the 16 folded ciphertexts packed and switched by the oracle (pack :198-241,
    # :1074-1081) must give the device's packed ciphertext and response
    gsw = M.regev_to_simple_gsw(cv, v, po.t_conv, s.ell, po.nu2)
    neg = M.pack_fold_neg(gsw, s.ell, po.nu2)
    v_ct = np.zeros((s.trials, 2, N), dtype=np.uint64)
    shas = []
    for t in range(s.trials):
        db_t = M.pack_gen_db_trial(po, out_n, seed, t)
        got_t = srv.read_acc(t)
        want_t = M.sweep_dim1(db_t, re, s.dim0, s.num_per)
        del db_t
        assert_eq(got_t, want_t, f"config 5, trial {t}: first-dimension accumulators")
        if t in (0, 7, 15):
            shas.append(f"{t}:{sha(got_t)}")
        v_ct[t] = M.fold_dim1(M.from_ntt(want_t), gsw, neg, s.ell, po.nu2)
    record(request, f"config 5: first-dimension accumulators of all {s.trials} trials bit-exact in full (sha256 {' '.join(shas)})")
    want_packed = M.pack(v_ct, vw, out_n, po.t_conv)
    praw = M.from_ntt(want_packed)
    want_resp = np.stack([np.array([M.rescale(int(x) % M.Q, M.Q, s.qprime if r == 0 else 4 * po.p_db) for x in praw[r].ravel()], dtype=np.uint64).reshape(praw[r].shape)
                          for r in range(out_n + 1)])
    assert_eq(packed, want_packed, "config 5: packed ciphertext")
    assert_eq(resp, want_resp, "config 5: response")
    record(request, f"config 5: all-trials: yes (streamed one trial at a time) -- packed ciphertext and response bit-exact over all {s.trials} trials; sha256 response {sha(resp)}")
    srv.close()

"""GPU parity of the SpiralPack / SpiralStreamPack path (reference src/testing.cpp) against the oracle, through the
C ABI: the two function seams, the resident server (packed ciphertext and response bit-exact, response decodes to
the out_n x out_n items)).  BASELINE.json config 5 at full size: tests/test_gpu_fullsize.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
N = 2048


@pytest.fixture(scope="module")
def sa():
    # torch first: it ships its own HIP runtime and the two must not be initialised in the opposite order
    import torch

    torch.cuda.is_available()
    import spiral_amd

    assert spiral_amd.lib().spiral_gpu_device_count() > 0
    return spiral_amd


def rand_ntt(rng, O, shape):
    return np.stack([rng.integers(0, m, size=shape + (N,), dtype=np.uint64) for m in (O.P, O.B)], axis=-2)


def assert_eq(got, exp, what):
    if not (got.shape == exp.shape and (got == exp).all()):
        bad = np.argwhere(got != exp)
        raise AssertionError(f"{what}: {len(bad)} of {got.size} words differ, first at {bad[:5].tolist()}")


@pytest.mark.parametrize("dim0,num_per", [(4, 2), (8, 32), (64, 64), (2, 128), (16, 128), (512, 64), (16, 2), (256, 8), (512, 32)])  # dim0 % 16 == 0: packed layout (wide, per-lane records, staged records)
def test_sweep_dim1(sa, oracle, dim0, num_per):
    O = oracle
    rng = np.random.default_rng(dim0 * 1000 + num_per)
    cts = rand_ntt(rng, O, (dim0, 2))
    re = O.reorient_dim1(cts, dim0, 1)
    db = O.fill_db_random(dim0 + num_per, dim0 * num_per * N)
    assert_eq(sa.fastMultiplyQueryByDatabaseDim1(db, re, dim0, num_per), O.sweep_dim1(db, re, dim0, num_per), "fastMultiplyQueryByDatabaseDim1")


@pytest.mark.parametrize("out_n,t_conv", [(2, 4), (3, 4), (4, 8), (2, 56)])
def test_pack_seam(sa, oracle, out_n, t_conv):
    O = oracle
    rng = np.random.default_rng(out_n * 100 + t_conv)
    v_ct = rng.integers(0, O.Q, size=(out_n * out_n, 2, N), dtype=np.uint64)
    v_ct[0, 0, :3] = [0, O.Q - 1, 1]
    v_w = rand_ntt(rng, O, (out_n, out_n + 1, t_conv))
    assert_eq(sa.pack(out_n, t_conv, v_ct, v_w), O.pack(v_ct, v_w, out_n, t_conv), "pack")


def test_pack_seam_largest_sums(sa, oracle):
    """out_n = 16, t_conv = 56 with every key residue m - 1: 896 products of ~2^56 per output word, far more than a u64
    holds unreduced -- the reference reduces after each r's t_conv terms (multiply, src/poly.cpp:62) and adds mod m"""
    O = oracle
    out_n, t_conv = 16, 56
    rng = np.random.default_rng(1656)
    v_ct = rng.integers(0, O.Q, size=(out_n * out_n, 2, N), dtype=np.uint64)
    v_w = np.zeros((out_n, out_n + 1, t_conv, 2, N), dtype=np.uint64)
    v_w[..., 0, :] = O.P - 1
    v_w[..., 1, :] = O.B - 1
    assert_eq(sa.pack(out_n, t_conv, v_ct, v_w), O.pack(v_ct, v_w, out_n, t_conv), "pack, out_n = 16, t_conv = 56, keys m - 1")


@pytest.mark.parametrize(
    "nu1,nu2,out_n,kw",
    [
        (6, 2, 2, {}),
        (5, 2, 3, dict(t_gsw=4)),
        (4, 6, 2, dict(t_gsw=4)),  # num_per = 64: the fast sweep path
        (3, 2, 2, dict(t_gsw=5, t_exp=2, qprime_bits=19, direct_upload=1)),
        (3, 2, 12, dict(t_gsw=3, t_conv=56, t_exp=56, qprime_bits=31, p_db=524288, direct_upload=1)),  # n = 12: the "movie" set's shape of parameters
    ],
)
@pytest.mark.parametrize("fold_pair", [1, 0, 2])  # 1: the pair form of foldCiphertextsDim1 (LD_PDIFF); 0: the reference's two products;
                                                   # 2: the pair form with every digit launch through the two-digits-per-workgroup kernel (option fwd2 = 1; by default only from 8192 transforms)
def test_pack_server_matches_oracle(sa, oracle, nu1, nu2, out_n, kw, fold_pair, request):
    O = oracle
    request.addfinalizer(lambda old=(sa.get_option("fold_pair"), sa.get_option("fwd2")): (sa.set_option("fold_pair", old[0]), sa.set_option("fwd2", old[1])))
    sa.set_option("fold_pair", 1 if fold_pair else 0)
    sa.set_option("fwd2", 1 if fold_pair == 2 else -1)
    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    s = O.pack_shape_of(po, out_n)
    g = sa.get_pack_shape(pg, out_n)
    for f, _ in g._fields_:
        assert getattr(g, f) == getattr(s, f), f
    db = O.pack_gen_db(po, out_n, 31)
    cl = O.PackClient(po, out_n, seed=8)
    wl, wr, v, vw = cl.pub_params()
    total = s.dim0 * s.num_per
    srv = sa.PackServer(pg, out_n)
    srv.gen_db(31)  # on-device generation must equal the oracle's database
    srv.set_pub_params(wl, wr, v, vw)
    for idx in (77 % total, total - 1):
        q = cl.query(idx)
        resp, packed, us = srv.answer(q)
        exp_resp, exp_packed = O.pack_answer(po, out_n, q, wl, wr, v, vw, db)
        assert_eq(packed, exp_packed, "packed ciphertext")
        assert_eq(resp, exp_resp, "response")
        assert_eq(cl.decode(resp), O.pack_db_item(po, out_n, 31, idx), "decoded items")
    # the same database uploaded in the reference's convertDb layout
    srv2 = sa.PackServer(pg, out_n)
    for t in range(s.trials):
        srv2.load_db(t, db[t])
    srv2.set_pub_params(wl, wr, v, vw)
    q = cl.query(3 % total)
    assert_eq(srv2.answer(q)[0], O.pack_answer(po, out_n, q, wl, wr, v, vw, db)[0], "response with uploaded database")
    srv.close()
    srv2.close()


@pytest.mark.parametrize("nu1,nu2,out_n,kw", [(4, 2, 2, dict(t_gsw=4)), (3, 2, 5, dict(t_gsw=3, t_conv=56, t_exp=56, qprime_bits=31, p_db=524288, direct_upload=1))])
def test_pack_response_wire_form(sa, oracle, nu1, nu2, out_n, kw):
    """SpiralPack response bit-packed on the device: n polynomials at q' bits, n^2 at log2(4p) bits == the oracle's wire bytes"""
    O = oracle
    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    cl = O.PackClient(po, out_n, seed=4)
    srv = sa.PackServer(pg, out_n)
    srv.gen_db(11)
    srv.set_pub_params(*cl.pub_params())
    s = O.pack_shape_of(po, out_n)
    for idx in (1, s.dim0 * s.num_per - 1):
        resp, _, _ = srv.answer(cl.query(idx), want_packed=False)
        wire = srv.read_response_wire()
        assert wire.size == O.response_wire_bytes(po, out_n)
        assert_eq(wire, O.response_to_wire(po, resp, out_n), "wire bytes")
        assert_eq(sa.response_from_wire(pg, wire, out_n), resp, "client unpack")
        assert_eq(cl.decode(sa.response_from_wire(pg, wire, out_n)), O.pack_db_item(po, out_n, 11, idx), "decoded from the wire form")
    srv.close()


@pytest.mark.parametrize("nu1,nu2,out_n,kw,cuts", [(4, 2, 2, dict(t_gsw=4), (0, 1, 3, 4)), (3, 3, 3, dict(t_gsw=3, t_conv=56, t_exp=56, qprime_bits=27, p_db=4096, direct_upload=1), (0, 5, 9)),
                                                   (4, 1, 4, dict(t_gsw=4), (0, 2, 4, 6, 8, 10, 12, 14, 16))])
def test_trial_sharded_servers_equal_one_server(sa, oracle, nu1, nu2, out_n, kw, cuts):
    """N GPUs split the out_n^2 trials (emulated on one device): every shard folds its own trials into its slice of the buffer an
    all-gather would fill, the root packs the gathered ciphertexts -- packed ciphertext and response == the oracle's whole answer"""
    import torch

    O = oracle
    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    s = O.pack_shape_of(po, out_n)
    db = O.pack_gen_db(po, out_n, 13)
    cl = O.PackClient(po, out_n, seed=2)
    pp = cl.pub_params()
    shards = [sa.PackServer(pg, out_n, 0, a, b) for a, b in zip(cuts[:-1], cuts[1:])]
    for k, sh in enumerate(shards):
        if k % 2 == 0:
            sh.gen_db(13)  # generated on the device with the global trial index
        else:
            for t in range(sh.trial0, sh.trial1):
                sh.load_db(t, db[t])
            with pytest.raises(RuntimeError):
                sh.load_db(sh.trial1 % s.trials if sh.trial1 % s.trials not in range(sh.trial0, sh.trial1) else (sh.trial0 - 1) % s.trials, db[0])
        sh.set_pub_params(*pp)
    if len(shards) > 1:
        with pytest.raises(RuntimeError):
            shards[0].answer(cl.query(0))
    gathered = torch.zeros(s.trials * 2 * N, dtype=torch.int64, device="cuda")
    total = s.dim0 * s.num_per
    for idx in (0, total - 1, total // 2 + 1):
        q = cl.query(idx)
        for sh in shards:
            sh.fold_trials(q, gathered.data_ptr() + sh.trial0 * 2 * N * 8)
        torch.cuda.synchronize()
        resp, packed = shards[0].pack_gathered(gathered.data_ptr(), want_packed=True)
        exp_resp, exp_packed = O.pack_answer(po, out_n, q, *pp, db)
        assert_eq(packed, exp_packed, f"packed ciphertext idx={idx}")
        assert_eq(resp, exp_resp, "response")
        if "p_db" not in kw:
            assert_eq(cl.decode(resp), O.pack_db_item(po, out_n, 13, idx), "decoded items")
    for sh in shards:
        sh.close()


def _random_pack_sets(count, seed):
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        nu1, nu2, out_n = int(rng.integers(2, 6)), int(rng.integers(0, 5)), int(rng.choice([2, 3, 4, 5, 7, 8, 12]))
        kw = dict(t_gsw=int(rng.integers(2, 11)), t_conv=int(rng.choice([2, 3, 4, 8, 16, 56])), t_exp=int(rng.choice([2, 4, 5, 8, 16, 56])),
                  qprime_bits=int(rng.integers(14, 37)), p_db=int(rng.choice([2, 256, 4096, 65536, 1 << 19])), direct_upload=int(rng.integers(0, 2)))
        if out_n * out_n * (1 << (nu1 + nu2)) > 16384:  # the oracle builds out_n^2 trial databases
            continue
        if nu2 < 1 or (not kw["direct_upload"] and kw["t_gsw"] * nu2 > (1 << nu1)):  # the packing path needs a fold and a stop round (src/testing.cpp:957-965)
            continue
        out.append((nu1, nu2, out_n, kw))
    return out


_N_FUZZ = int(__import__("os").environ.get("SPIRAL_FUZZ_SETS", "12"))  # a soak run sets it to hundreds


@pytest.mark.parametrize("nu1,nu2,out_n,kw", _random_pack_sets(_N_FUZZ, 7), ids=[f"set{i}" for i in range(_N_FUZZ)])
def test_random_pack_sets_bit_exact(sa, oracle, nu1, nu2, out_n, kw):
    """a seeded draw of SpiralPack / SpiralStreamPack parameter sets (odd gadget dimensions and output sizes, every q' width):
    packed ciphertext and switched response == the oracle's, word for word"""
    O = oracle
    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    s = O.pack_shape_of(po, out_n)
    db = O.pack_gen_db(po, out_n, 5)
    cl = O.PackClient(po, out_n, seed=nu1 + 10 * nu2)
    wl, wr, v, vw = cl.pub_params()
    srv = sa.PackServer(pg, out_n)
    srv.gen_db(5)
    srv.set_pub_params(wl, wr, v, vw)
    total = s.dim0 * s.num_per
    for idx in (0, total - 1):
        q = cl.query(idx)
        resp, packed, _ = srv.answer(q)
        exp_resp, exp_packed = O.pack_answer(po, out_n, q, wl, wr, v, vw, db)
        assert_eq(packed, exp_packed, f"packed ciphertext idx={idx} params {nu1},{nu2},{out_n},{kw}")
        assert_eq(resp, exp_resp, "response")
    srv.close()


def test_pack_raw_ingest(sa, oracle):
    """raw ingest of the trial databases (1 x 1 plaintexts, src/testing.cpp:845-869 + convertDb :316-340 on the device):
    same answers as the device-generated and as the uploaded database"""
    O = oracle
    kw = dict(t_gsw=4)
    nu1, nu2, out_n = 5, 2, 2
    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    s = O.pack_shape_of(po, out_n)
    total = s.dim0 * s.num_per
    cl = O.PackClient(po, out_n, seed=8)
    wl, wr, v, vw = cl.pub_params()
    db = O.pack_gen_db(po, out_n, 31)
    srv = sa.PackServer(pg, out_n)
    all_pts = np.stack([O.pack_db_item(po, out_n, 31, i).reshape(s.trials, N) for i in range(total)])  # [item][trial][N]
    for t in range(s.trials):
        srv.load_db_items(t, O.pack_items(all_pts[:, t], 8), 8)
    srv.set_pub_params(wl, wr, v, vw)
    q = cl.query(total - 1)
    resp, packed, _ = srv.answer(q)
    want_resp, want_packed = O.pack_answer(po, out_n, q, wl, wr, v, vw, db)
    assert_eq(packed, want_packed, "packed ciphertext from the ingested database")
    assert_eq(resp, want_resp, "response from the ingested database")
    srv.close()

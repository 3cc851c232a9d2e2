"""GPU parity tests: every seam of the C ABI against the oracle on the same seeded inputs.
Bit-exact: the whole path is unsigned integer arithmetic (raw-domain values compared verbatim,
NTT-domain values compared as canonical residues, SURVEY.md section 8c hazard 1)."""
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

    assert spiral_amd.lib().spiral_gpu_device_count() > 0, "GPU tests need a device"
    return spiral_amd


def rand_ntt(rng, O, shape):
    """random canonical NTT-form polys, shape + (2, N)"""
    return np.stack([rng.integers(0, m, size=shape + (N,), dtype=np.uint64) for m in (O.P, O.B)], axis=-2)


def canon(O, a):
    a = a.copy()
    a[..., 0, :] %= O.P
    a[..., 1, :] %= O.B
    return a


def assert_eq(got, exp, what):
    if not (got.shape == exp.shape and (got == exp).all()):
        bad = np.argwhere(got != exp)
        raise AssertionError(f"{what}: {len(bad)} of {got.size} words differ, first at {bad[:5].tolist()}: got {got[tuple(bad[0])]}, exp {exp[tuple(bad[0])]}")


# ---- L1 / L2 -----------------------------------------------------------------------------------------------
def test_ntt_forward_inverse(sa, oracle):
    O = oracle
    rng = np.random.default_rng(10)
    x = rand_ntt(rng, O, (5,))
    x[0, :, :4] = 0
    x[1, 0, :] = O.P - 1
    x[1, 1, :] = O.B - 1
    x[2] *= 3  # lazy inputs < 4m are legal for ntt_forward (src/core.cpp:274)
    f = sa.ntt_forward(x)
    assert_eq(f, O.ntt_forward(x), "ntt_forward")
    xi = rand_ntt(rng, O, (9,))
    # the unscaled inverse stages double their sums per stage: the patterns that drive every register of every pass to its bound
    top = np.array([O.P - 1, O.B - 1], dtype=np.uint64)[:, None]
    xi[0] = top                                    # every sum maximal (16 m - 16 before a pass's reduction)
    xi[1] = top * (np.arange(N) % 2 == 0)          # maximal differences in the first stage
    xi[2] = top * (np.arange(N) % 2 == 1)
    xi[3] = top * (np.arange(N) < N // 2)          # ... in the last
    xi[4] = top * ((np.arange(N) // 8) % 2 == 0)   # ... across the first exchange
    xi[5] = top * ((np.arange(N) // 64) % 2 == 1)
    assert_eq(sa.ntt_inverse(xi), O.ntt_inverse(xi), "ntt_inverse")
    assert_eq(sa.from_ntt(xi), O.from_ntt(xi), "from_ntt on the extreme patterns (lazy CRT lift)")
    assert_eq(sa.ntt_inverse(f), canon(O, x), "inverse(forward)")


def test_to_from_ntt(sa, oracle):
    O = oracle
    rng = np.random.default_rng(11)
    A = rng.integers(0, O.Q, size=(3, 6, N), dtype=np.uint64)
    A[0, 0, :5] = [0, 1, O.Q - 1, O.Q, O.P]
    An = sa.to_ntt(A)
    assert_eq(An, O.to_ntt(A), "to_ntt")
    assert_eq(sa.from_ntt(An), A % O.Q, "do_MatPol_test round trip")  # src/spiral.cpp:1181
    assert_eq(sa.from_ntt(An), O.from_ntt(An), "from_ntt")
    d = rng.integers(0, 1 << 29, size=(4, N), dtype=np.uint64)
    assert_eq(sa.to_ntt_no_reduce(d), O.to_ntt(d, reduce=False), "to_ntt_no_reduce")


def test_multiply_add_mul_by_const(sa, oracle):
    O = oracle
    rng = np.random.default_rng(12)
    for rs, ms, cs in [(2, 8, 1), (3, 8, 2), (3, 56, 1), (1, 1, 1)]:
        a, b = rand_ntt(rng, O, (rs, ms)), rand_ntt(rng, O, (ms, cs))
        assert_eq(sa.multiply(a, b), O.multiply(a, b), f"multiply {rs}x{ms}x{cs}")
    a, b = rand_ntt(rng, O, (3, 2)), rand_ntt(rng, O, (3, 2))
    assert_eq(sa.add(a, b), O.add(a, b), "add")
    s = rand_ntt(rng, O, ())
    assert_eq(sa.mul_by_const(s, a), O.mul_by_const(s, a), "mul_by_const")


def test_raw_domain_ops(sa, oracle):
    O = oracle
    rng = np.random.default_rng(13)
    a = rng.integers(0, O.Q, size=(2, N), dtype=np.uint64)
    a[0, :3] = [0, O.Q - 1, 1]
    for r in range(0, 11):
        t = (N >> r) + 1
        assert_eq(sa.automorph(a, t), O.automorph(a, t), f"automorph t={t}")
    assert_eq(sa.invert(a), O.invert(a), "invert")
    for t in (2, 4, 5, 8, 10, 16, 56):
        v = rng.integers(0, O.Q, size=(1, 2, N), dtype=np.uint64)
        v[0, 0, 0] = O.Q
        assert_eq(sa.gadget_invert(v, t, 1), O.gadget_invert(v, t, 1), f"gadget_invert t={t}")
    v = rng.integers(0, O.Q, size=(2, 3, N), dtype=np.uint64)
    assert_eq(sa.gadget_invert(v, 8, 2), O.gadget_invert(v, 8, 2), "gadget_invert rdim=2")


def test_rescale(sa, oracle):
    O = oracle
    rng = np.random.default_rng(14)
    a = rng.integers(0, O.Q, size=4096, dtype=np.uint64)
    a[:8] = [0, 1, O.Q // 2 - 1, O.Q // 2, O.Q // 2 + 1, O.Q - 1, O.Q, 12345]
    for out_mod in (786433, 1024, 12289, 68718428161, 4 * 32768):
        exp = np.array([O.rescale(int(x) % O.Q, O.Q, out_mod) for x in a], dtype=np.uint64)
        assert_eq(sa.getRescaled(a, O.Q, out_mod), exp, f"getRescaled -> {out_mod}")


# ---- L5 seams -------------------------------------------------------------------------------------------------
# (3,0) (4,1): fewer than 8 columns, per-lane query records; (3,2) (7,3) (9,4): 8..32 columns, records staged in LDS, the waves of
# a workgroup split the j range ((7,3): one group of 8 j per wave, (9,4): 4 per wave); (3,5) (5,6): 64+ columns; (2,1) (1,3)
# (2,7): dim0 < 8, plain layout
@pytest.mark.parametrize("nu1,nu2", [(2, 1), (1, 3), (3, 5), (5, 6), (2, 7), (3, 0), (4, 1), (3, 2), (7, 3), (9, 4)])
def test_multiply_query_by_database(sa, oracle, nu1, nu2):
    O = oracle
    dim0, num_per = 1 << nu1, 1 << nu2
    rng = np.random.default_rng(20 + nu1 * 16 + nu2)
    cts = rand_ntt(rng, O, (dim0, 3, 2))
    re = O.reorient_ciphertexts(cts)
    db = O.fill_db_random(nu1 * 100 + nu2, dim0 * num_per * 4 * N)
    assert_eq(sa.multiplyQueryByDatabase(re, db, dim0, num_per), O.multiply_query_by_database(re, db, dim0, num_per), "sweep")


# the batched sweep on the matrix cores (csrc/sweep_mfma.hip): num_per >= 64, dim0 a multiple of 64.  (6,6): one 128-term piece per prime and
# one workgroup column group; (7,6): two pieces; (6,7): two column groups per z; n = 1 .. 8 queries = 1 .. 6 column tiles, the last one part
# empty for n = 3, 5, 6, 7.  (5,6) and (6,5) fall back to the vector ALU (first dimension / columns too small) through the same entry point.
@pytest.mark.parametrize("nu1,nu2,n", [(6, 6, 1), (6, 6, 2), (6, 6, 3), (6, 6, 4), (6, 6, 5), (6, 6, 6), (6, 6, 7), (6, 6, 8), (7, 6, 4), (6, 7, 8), (8, 6, 5),
                                       (5, 6, 3), (6, 5, 4)])
def test_multiply_queries_by_database(sa, oracle, nu1, nu2, n):
    """n queries against one pass over the database == the oracle's multiplyQueryByDatabase (src/spiral.cpp:628-999) of every query"""
    O = oracle
    dim0, num_per = 1 << nu1, 1 << nu2
    rng = np.random.default_rng(900 + nu1 * 64 + nu2 * 8 + n)
    res = [O.reorient_ciphertexts(rand_ntt(rng, O, (dim0, 3, 2))) for _ in range(n)]
    db = O.fill_db_random(nu1 * 100 + nu2 + 7, dim0 * num_per * 4 * N)
    got = sa.multiplyQueriesByDatabase(res, db, dim0, num_per)
    for b, re in enumerate(res):
        assert_eq(got[b], O.multiply_query_by_database(re, db, dim0, num_per), f"query {b} of {n}")


@pytest.mark.parametrize("vq,vd", [("max", "max"), ("limb-", "limb-"), ("limb-", "limb+"), ("wrap", "max"), ("zero", "max")])
def test_matrix_core_sweep_extremes(sa, oracle, vq, vd):
    """constant operands at the edges of the limb decomposition, dim0 = 256 (K = 512 terms per sum): m - 1; the values whose three signed limb
    bytes are all -128 (top nibble 15) / all +127 (top nibble 14): the largest limb products of either sign; the first residue mod p stored as
    a - p (2^28 - 0x808080; as a query value its top limb is 16); zero.  Every output is 2 dim0 vq vd mod m."""
    O = oracle
    dim0, num_per, n = 256, 64, 3
    val = {"max": (O.P - 1, O.B - 1), "limb-": ((15 << 24) - 0x808080,) * 2, "limb+": ((14 << 24) - 0x808080 + 0xFFFFFF,) * 2,
           "wrap": ((1 << 28) - 0x808080, O.B - 1), "zero": (0, 0)}
    (qp, qb), (dp, db_) = val[vq], val[vd]
    assert max(qp, dp) < O.P and max(qb, db_) < O.B
    cts = np.zeros((dim0, 3, 2, 2, N), dtype=np.uint64)
    cts[..., 0, :] = qp
    cts[..., 1, :] = qb
    re = O.reorient_ciphertexts(cts)
    db = np.full(dim0 * num_per * 4 * N, dp | (db_ << 32), dtype=np.uint64)
    got = sa.multiplyQueriesByDatabase([re] * n, db, dim0, num_per)
    assert (got[..., 0, :] == (2 * dim0 * qp * dp) % O.P).all() and (got[..., 1, :] == (2 * dim0 * qb * db_) % O.B).all()


def test_sweep_accumulator_extremes(sa, oracle):
    """all operands m-1: the largest partial sums, exercises the 256-term reduction rule with dim0 = 256"""
    O = oracle
    dim0, num_per = 256, 32
    cts = np.zeros((dim0, 3, 2, 2, N), dtype=np.uint64)
    cts[..., 0, :] = O.P - 1
    cts[..., 1, :] = O.B - 1
    re = O.reorient_ciphertexts(cts)
    db = np.full(dim0 * num_per * 4 * N, (O.P - 1) | ((O.B - 1) << 32), dtype=np.uint64)
    got = sa.multiplyQueryByDatabase(re, db, dim0, num_per)
    assert (got[..., 0, :] == (2 * dim0 * (O.P - 1) ** 2) % O.P).all() and (got[..., 1, :] == (2 * dim0 * (O.B - 1) ** 2) % O.B).all()


@pytest.mark.parametrize("t_gsw", [4, 5, 8, 10])
def test_split_and_crt(sa, oracle, t_gsw):
    O = oracle
    rng = np.random.default_rng(30 + t_gsw)
    raw = rng.integers(0, O.Q, size=(3, 3, 2, N), dtype=np.uint64)
    bits = 56 // t_gsw + 1
    raw[0, 0, 0, :6] = [0, 1, O.Q - 1, 1 << (bits - 1), (1 << (bits - 1)) + 1, (1 << bits) - 1]
    assert_eq(sa.split_and_crt(raw, t_gsw), O.split_and_crt(raw, t_gsw), f"split_and_crt t={t_gsw}")


@pytest.mark.parametrize("num_per,t_gsw", [(1, 8), (4, 8), (2, 4), (2, 10)])
def test_fold_one_further_dimension(sa, oracle, num_per, t_gsw):
    O = oracle
    rng = np.random.default_rng(40 + num_per + t_gsw)
    m2 = 3 * t_gsw
    cts = rng.integers(0, O.Q, size=(2 * num_per, 3, 2, N), dtype=np.uint64)
    q = rand_ntt(rng, O, (3, m2))
    qn = rand_ntt(rng, O, (3, m2))
    q_re, qn_re = np.zeros(N * 3 * m2, dtype=np.uint64), np.zeros(N * 3 * m2, dtype=np.uint64)
    import ctypes as C

    O.lib().orc_reorient_Q(O._p(q_re), O._p(q), C.c_uint32(m2))
    O.lib().orc_reorient_Q(O._p(qn_re), O._p(qn), C.c_uint32(m2))
    exp = cts.copy()
    O.lib().orc_fold_one_further_dimension(O._p(exp), C.c_size_t(num_per), O._p(q_re), O._p(qn_re), C.c_uint32(t_gsw))
    got = sa.foldOneFurtherDimension(cts, num_per, q_re, qn_re, t_gsw)
    assert_eq(got, exp[:num_per], "foldOneFurtherDimension")


@pytest.mark.parametrize("g,t_exp,t_right,stopround,max_bits", [(3, 8, 56, 0, 0), (4, 8, 56, 2, 3), (5, 2, 56, 4, 16), (4, 16, 8, 0, 0)])
def test_expand_improved(sa, oracle, g, t_exp, t_right, stopround, max_bits):
    O = oracle
    rng = np.random.default_rng(50 + g)
    n_right = stopround + 1 if stopround else g
    cv = np.zeros((1 << g, 2, 2, N), dtype=np.uint64)
    cv[0] = rand_ntt(rng, O, (2,))
    wl = rand_ntt(rng, O, (g, 2, t_exp))
    wr = rand_ntt(rng, O, (n_right, 2, t_right))
    exp = O.expand_improved(cv, g, t_exp, wl, t_right, wr, n_right, max_bits, stopround)
    got = sa.expandImproved(cv, g, t_exp, wl, wr, t_right, n_right, max_bits, stopround)
    # odd slots the reference never processes nor reads again (created in round `stopround` or later but past
    # max_bits, src/spiral.cpp:1701-1702 + reorderFromStopround :2027) are dead: compare the live ones
    live = np.ones(1 << g, dtype=bool)
    if stopround:
        for i in range(1 << g):
            if i & 1 and (i >= (1 << (stopround + 1)) or (i >= (1 << stopround) and i // 2 > max_bits)):
                live[i] = False
    assert_eq(got[live], exp[live], "expandImproved")


def test_scal_to_mat_and_regev_to_gsw(sa, oracle):
    O = oracle
    rng = np.random.default_rng(60)
    for t_conv, ell in [(4, 8), (4, 4), (8, 5), (56, 4)]:
        cv = rand_ntt(rng, O, (ell, 2))
        w, v = rand_ntt(rng, O, (3, 2 * t_conv)), rand_ntt(rng, O, (3, 2 * t_conv))
        assert_eq(sa.scalToMat(t_conv, cv[0], w), O.scal_to_mat(cv[0], w, t_conv), f"scalToMat t_conv={t_conv}")
        assert_eq(sa.regevToGSW(t_conv, ell, cv, w, v), O.regev_to_gsw(cv, w, v, t_conv, ell), f"regevToGSW t_conv={t_conv} ell={ell}")


# ---- resident server, stage by stage and end to end -----------------------------------------------------------------
CONFIGS = [
    (2, 1, {}),  # stopround == 0
    (4, 2, dict(t_gsw=4)),  # stopround > 0
    (3, 3, dict(t_gsw=8)),
    (6, 2, {}),  # stopround with dim0 >> ell*nu2
    (2, 2, dict(t_gsw=5, t_exp=2, qprime_bits=19, direct_upload=1)),  # direct upload (SpiralStream-style)
    (2, 6, dict(t_gsw=8)),  # nic >= 64 but dim0 < 8: plain database layout, wide accumulator
    (3, 5, dict(t_gsw=8)),  # nic >= 64 and dim0 % 8 == 0: the packed 7-byte database layout and the fast sweep
]


@pytest.mark.parametrize("nu1,nu2,kw", CONFIGS)
def test_server_stages_match_oracle(sa, oracle, nu1, nu2, kw):
    O = oracle
    from spiral_amd import server as SV

    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    s = O.shape_of(po)
    total = s.dim0 * s.num_per
    idx = 1234 % total
    db = O.gen_db(po, 77)
    cl = O.Client(po, seed=5)
    wl, wr, w, v = cl.pub_params()
    q = cl.query(idx)

    srv = sa.Server(pg)
    srv.keep_cts(True)
    srv.load_db(db)
    srv.set_pub_params(wl, wr, w, v)
    srv.set_query(q)

    cv = O.stage_expand(po, q, wl, wr)
    srv.expand()
    assert_eq(srv.read(SV.BUF_EXPANDED), cv, "expanded ciphertexts")
    cts, gsw = O.stage_convert(po, cv, w, v)
    srv.convert()
    assert_eq(srv.read(SV.BUF_CTS), cts, "scalToMat outputs (expansionLocals.cts)")
    assert_eq(srv.read(SV.BUF_GSW), gsw, "regevToGSW outputs")
    raw = O.stage_first_dim(po, cts, db)
    srv.first_dim()
    srv.lift()
    assert_eq(srv.read(SV.BUF_RAW), raw, "first dimension (sweep + INTT + CRT)")
    fin = O.stage_fold(po, raw, gsw)
    srv.fold()
    assert_eq(srv.read(SV.BUF_FINAL), fin, "folded ciphertext")
    srv.finish()
    resp = O.stage_rescale(po, fin)
    assert_eq(srv.read(SV.BUF_RESPONSE), resp, "response")
    # the reference's own functional check: Is correct?
    assert_eq(cl.decode(srv.read(SV.BUF_RESPONSE)), O.db_item(po, 77, idx), "decoded plaintext")
    srv.close()


@pytest.mark.parametrize("nu1,nu2,kw", [(4, 3, dict(t_gsw=4)), (2, 2, dict(t_gsw=5, t_exp=2, qprime_bits=19, direct_upload=1))])
def test_answer_and_device_db(sa, oracle, nu1, nu2, kw):
    """answer() in one call with the database generated ON the device (same seeded coefficients as the oracle's)"""
    O = oracle
    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    s = O.shape_of(po)
    cl = O.Client(po, seed=9)
    wl, wr, w, v = cl.pub_params()
    srv = sa.Server(pg)
    srv.gen_db(4321)
    srv.set_pub_params(wl, wr, w, v)
    db = O.gen_db(po, 4321)
    for idx in (0, s.dim0 * s.num_per - 1, 7):
        q = cl.query(idx)
        fin, resp, us = srv.answer(q)
        assert_eq(fin, O.answer(po, q, wl, wr, w, v, db), "answer")
        assert_eq(cl.decode(resp), O.db_item(po, 4321, idx), "decoded plaintext")
        assert us["total_us"] > 0
    srv.close()


def test_load_db_in_several_staging_passes(sa, oracle, opts):
    """load_db stages the reference-layout database a few z slabs at a time; every pass must land its slabs at the
    right (thread-transposed) positions of the packed device layout: answer == oracle, and == the device-generated DB"""
    O = oracle
    from spiral_amd import server as SV

    kw = dict(t_gsw=8)
    po, pg = O.make_params(4, 5, **kw), sa.make_params(4, 5, **kw)
    cl = O.Client(po, seed=11)
    wl, wr, w, v = cl.pub_params()
    db = O.gen_db(po, 99)
    q = cl.query(333)
    want = O.answer(po, q, wl, wr, w, v, db)
    opts(db_stage_bytes=48 * 16 * 32 * 4 * 8)  # 48 slabs per pass: 43 passes, the last one short
    srv = sa.Server(pg)
    srv.load_db(db)
    srv.set_pub_params(wl, wr, w, v)
    fin, resp, _ = srv.answer(q)
    assert_eq(fin, want, "answer from a database loaded in 43 passes")
    acc_loaded = srv.read(SV.BUF_ACC)
    srv.gen_db(99)
    fin2, _, _ = srv.answer(q)
    assert_eq(fin2, want, "answer from the device-generated database")
    assert_eq(srv.read(SV.BUF_ACC), acc_loaded, "accumulators: loaded vs generated database")
    srv.close()


@pytest.mark.parametrize("env,t_gsw", [
    ({}, 8),                                    # default: pair form, unchained (lift launch + LD_SDIFF launch + product with addend)
    (dict(fold_pair=0), 8),                     # the reference's two-product form Q_neg G^-1(L) + Q G^-1(H), lift chained (fold_chain_kernel)
    (dict(fold_pair=0, fold_blocks=0), 8),        # one block per polynomial (all digits)
    (dict(fold_pair=0, fold_blocks=1000000), 8),  # one block per (polynomial, digit)
    (dict(fold_pair=0, fold_blocks=300), 8),      # mixed chunk sizes
    (dict(fold_chain=0), 8),                    # two-product form as separate lift + LD_SDIGIT launches
    ({}, 14),                                   # NO option: ell = 14 has (ell - 1) * bits = 65 >= 64, fold_pair_exact(14) is false, so the server
    ({}, 17),                                   # itself falls back to the two-product form (kernels.h fold_pair_exact); likewise ell = 17
    (dict(fwd2=1), 8),                          # every digit launch through the two-digits-per-workgroup kernel (default only from 8192 transforms)
    (dict(fwd2=0), 8),
    (dict(fwd2=1), 7),                          # odd digit counts: the last job of a source carries one digit (the fold's LD_SDIFF and the conversion's LD_DIGIT
    (dict(fwd2=1), 9),                          # through the two-digit kernel)
])
def test_fold_chain_schedules(sa, oracle, env, t_gsw, opts):
    """the fold's forms (library options taken when the server is created -- spiral_gpu_set_option -- or chosen by the server from the gadget
    dimension) and the two digit-transform kernels all give the oracle's answer"""
    O = oracle
    opts(**env)
    kw = dict(t_gsw=t_gsw)  # (t_gsw = 4 is bit-exact too but too noisy to decode at nu2 = 6)
    po, pg = O.make_params(3, 6, **kw), sa.make_params(3, 6, **kw)
    cl = O.Client(po, seed=17)
    wl, wr, w, v = cl.pub_params()
    srv = sa.Server(pg)
    srv.gen_db(5)
    srv.set_pub_params(wl, wr, w, v)
    db = O.gen_db(po, 5)
    q = cl.query(301)
    want = O.answer(po, q, wl, wr, w, v, db)
    fin, resp, _ = srv.answer(q)                  # stage API: lift, then fold from the lifted ciphertexts
    assert_eq(fin, want, f"answer {env}")
    srv.use_graphs(True)
    srv.set_query(q)
    srv.run_query()                               # lift chained into round 0
    srv.sync()
    from spiral_amd import server as SV
    assert_eq(srv.read(SV.BUF_FINAL), want, f"run_query {env}")
    assert_eq(cl.decode(srv.read(SV.BUF_RESPONSE)), O.db_item(po, 5, 301), "decoded plaintext")
    srv.close()


@pytest.mark.parametrize("graphs,overlap", [(True, 0), (False, 0), (True, 2), (False, 2)])
def test_graph_replay_matches_eager(sa, oracle, graphs, overlap):
    """run_pre / run_post captured into hipGraphs and replayed for several queries == the eager stages; with the split schedule
    (overlap 2) the GSW side of the query runs on the side stream and the fold joins it"""
    O = oracle
    from spiral_amd import server as SV

    kw = dict(t_gsw=4)
    po, pg = O.make_params(4, 3, **kw), sa.make_params(4, 3, **kw)
    cl = O.Client(po, seed=21)
    wl, wr, w, v = cl.pub_params()
    srv = sa.Server(pg)
    srv.gen_db(8)
    srv.set_pub_params(wl, wr, w, v)
    db = O.gen_db(po, 8)
    srv.use_graphs(graphs)
    srv.set_overlap(overlap)
    for idx in (3, 100, 127, 3):
        q = cl.query(idx)
        srv.set_query(q)
        srv.run_pre()
        srv.first_dim()
        srv.run_post()
        srv.sync()
        assert_eq(srv.read(SV.BUF_FINAL), O.answer(po, q, wl, wr, w, v, db), f"graph replay idx={idx}")
        assert_eq(cl.decode(srv.read(SV.BUF_RESPONSE)), O.db_item(po, 8, idx), "decoded plaintext")
    fin, resp, us = srv.answer(cl.query(5))  # answer() through the graphs too
    assert_eq(cl.decode(resp), O.db_item(po, 8, 5), "decoded plaintext via answer()")
    for idx in (9, 64):  # the whole query as one group
        q = cl.query(idx)
        srv.set_query(q)
        srv.run_query()
        srv.sync()
        assert_eq(srv.read(SV.BUF_FINAL), O.answer(po, q, wl, wr, w, v, db), f"run_query idx={idx}")
    srv.close()


@pytest.mark.parametrize("kw", [dict(t_gsw=4), dict(t_gsw=4, qprime_bits=27, p_db=32768, direct_upload=1), dict(t_gsw=5, qprime_bits=36, p_db=8388592),
                                dict(t_gsw=4, qprime_bits=14, p_db=4)])
def test_response_wire_form(sa, oracle, kw):
    """the response bit-packed on the device == the oracle's wire bytes of the same response; unpacked by the client half it is the
    response again and decodes to the item where the parameters allow"""
    O = oracle
    from spiral_amd import server as SV

    po, pg = O.make_params(4, 3, **kw), sa.make_params(4, 3, **kw)
    cl = O.Client(po, seed=3)
    srv = sa.Server(pg)
    srv.gen_db(9)
    srv.set_pub_params(*cl.pub_params())
    for idx in (0, 77, 127):
        fin, resp, _ = srv.answer(cl.query(idx))
        wire = srv.read_response_wire()
        assert wire.size == sa.response_wire_bytes(pg) and wire.size < resp.size * 8 // 2
        assert_eq(wire, O.response_to_wire(po, resp), f"wire bytes idx={idx}")
        assert_eq(sa.response_from_wire(pg, wire), resp, "client unpack")
        assert_eq(srv.read(SV.BUF_RESPONSE), resp, "response buffer untouched")
    srv.close()


def _random_parameter_sets(count, seed):
    """valid parameter sets the fixed cases do not visit: odd gadget dimensions, every q' width, tiny and large plaintext moduli, both
    query forms, expansions with and without a stop round"""
    rng = np.random.default_rng(seed)
    out = []
    while len(out) < count:
        nu1, nu2 = int(rng.integers(1, 7)), int(rng.integers(0, 5))
        kw = dict(t_gsw=int(rng.integers(2, 13)), t_conv=int(rng.choice([1, 2, 3, 4, 7, 8, 16, 28, 56])), t_exp=int(rng.choice([2, 3, 4, 5, 8, 16, 28, 56])),
                  t_exp_right=int(rng.choice([4, 8, 28, 56])), qprime_bits=int(rng.integers(14, 37)), p_db=int(rng.choice([2, 4, 256, 4096, 65536, 1 << 20])),
                  direct_upload=int(rng.integers(0, 2)))
        if not kw["direct_upload"] and (1 << nu1) + kw["t_gsw"] * nu2 > 2048:
            continue
        out.append((nu1, nu2, kw))
    return out


def _with_matrix_core_geometries(sets):
    """every third set of a draw moved to a geometry the matrix-core sweep covers (first dimension 64 or 128, at least 64 ciphertexts per slot), keeping its
    drawn gadget dimensions, moduli and query form: the limb-plane image and sweep_mfma_kernel then meet the same variety the vector-ALU sweep does"""
    out = []
    for i, (nu1, nu2, kw) in enumerate(sets):
        if i % 3 == 2:
            nu1, nu2 = ((6, 6), (7, 6), (6, 6))[(i // 3) % 3]
            if not kw["direct_upload"] and (1 << nu1) + kw["t_gsw"] * nu2 > 2048:
                kw = dict(kw, t_gsw=8)
        out.append((nu1, nu2, kw))
    return out


def _has_limb_form(nu1, nu2):
    return 6 <= nu1 <= 11 and nu2 >= 6  # (sweep_mfma.hip sweep_mfma_ok: first dimension a power of two in [64, 2048], num_per >= 64)


_N_FUZZ = int(__import__("os").environ.get("SPIRAL_FUZZ_SETS", "16"))  # a soak run sets it to hundreds


@pytest.mark.parametrize("nu1,nu2,kw", _random_parameter_sets(_N_FUZZ, 2024), ids=[f"set{i}" for i in range(_N_FUZZ)])
def test_random_parameter_sets_bit_exact(sa, oracle, nu1, nu2, kw):
    """a seeded draw of parameter sets: the folded ciphertext of the eager stages and of the whole-query graph == the oracle's, word for
    word (whether such a set decodes is the noise model's business, not the server's)"""
    O = oracle
    from spiral_amd import server as SV

    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    cl = O.Client(po, seed=nu1 * 31 + nu2)
    wl, wr, w, v = cl.pub_params()
    seed = 77 + nu1
    db = O.gen_db(po, seed)
    srv = sa.Server(pg)
    srv.gen_db(seed)
    srv.set_pub_params(wl, wr, w, v)
    total = 1 << (nu1 + nu2)
    for k, idx in enumerate((0, total - 1, total // 3)):
        q = cl.query(idx)
        want = O.answer(po, q, wl, wr, w, v, db)
        if k == 0:
            fin, resp, _ = srv.answer(q)
        else:
            srv.use_graphs(True)
            srv.set_query(q)
            srv.run_query()
            srv.sync()
            fin = srv.read(SV.BUF_FINAL)
        assert_eq(fin, want, f"final ciphertext, idx={idx}, params {nu1},{nu2},{kw}")
    srv.close()


_N_FUZZ_BATCH = max(6, _N_FUZZ // 4)


@pytest.mark.parametrize("nu1,nu2,kw", _with_matrix_core_geometries(_random_parameter_sets(_N_FUZZ_BATCH, 515)), ids=[f"set{i}" for i in range(_N_FUZZ_BATCH)])
def test_random_parameter_sets_batched(sa, oracle, nu1, nu2, kw):
    """another seeded draw of parameter sets through run_query_batch with 2-8 lanes (each lane its own client): every lane's folded ciphertext and
    response == the oracle's for its inputs -- odd gadget dimensions, both query forms, expansions with and without a stop round, tiny geometries
    whose sweep falls back to one launch per lane"""
    O = oracle
    from spiral_amd import server as SV

    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    n = 2 + (nu1 * 7 + nu2 * 3 + kw["t_gsw"]) % 7
    seed = 91 + nu2
    db = O.gen_db(po, seed)
    owner = sa.Server(pg)
    owner.gen_db(seed)
    lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(n - 1)]
    clients = [O.Client(po, seed=200 + 13 * b + nu1) for b in range(n)]
    pps = [cl.pub_params() for cl in clients]
    for srv, pp in zip(lanes, pps):
        srv.set_pub_params(*pp)
        srv.use_graphs(True)
    total = 1 << (nu1 + nu2)
    for rnd in range(2):
        idxs = [(rnd * 5 + 3 * b) % total for b in range(n)]
        qs = [cl.query(i) for cl, i in zip(clients, idxs)]
        for srv, q in zip(lanes, qs):
            srv.set_query(q)
        sa.run_query_batch(lanes)
        for b, (srv, pp, q) in enumerate(zip(lanes, pps, qs)):
            srv.sync()
            want = O.answer(po, q, pp[0], pp[1], pp[2], pp[3], db)
            assert_eq(srv.read(SV.BUF_FINAL), want, f"round {rnd} lane {b} of {n}: final ciphertext, params {nu1},{nu2},{kw}")
            assert_eq(srv.read(SV.BUF_RESPONSE), O.stage_rescale(po, want), f"round {rnd} lane {b} of {n}: response")
    # one image per server: where the batch swept on the matrix cores it converted the owner's image in place; a single query then sweeps the limb planes
    # (sweep_mfma_kernel<1>), and converting back restores the packed words exactly -- on this draw's geometry
    fmt = owner.db_format()
    if _has_limb_form(nu1, nu2) and n >= max(sa.get_option("sweep_mfma_min"), 1) and sa.get_option("sweep_mfma_min") and sa.get_option("one_image"):
        assert fmt == SV.DB_LIMBS, f"a batch of {n} at ({nu1}, {nu2}) should have converted the image"
    want0 = O.answer(po, qs[0], *pps[0], db)
    owner.run_query()
    owner.sync()
    assert_eq(owner.read(SV.BUF_FINAL), want0, f"single query on the image as the batch left it (format {fmt}), params {nu1},{nu2},{kw}")
    item = (7 * nu1 + nu2) % total
    before = owner.read_db_item(item)
    if fmt == SV.DB_LIMBS:
        owner.set_db_format(SV.DB_PACKED)
        assert owner.db_format() == SV.DB_PACKED
    assert_eq(owner.read_db_item(item), before, f"item {item} after converting the image back, params {nu1},{nu2},{kw}")
    owner.run_query()
    owner.sync()
    assert_eq(owner.read(SV.BUF_FINAL), want0, f"single query after converting the image back, params {nu1},{nu2},{kw}")
    for srv in lanes[1:]:
        srv.close()
    owner.close()


def test_two_query_lanes_share_one_database(sa, oracle):
    """share_db: a second server handle sweeps the first one's database image; queries of two clients in flight on two streams,
    interleaved, every answer bit-exact; loading through the lane is refused"""
    import torch

    O = oracle
    from spiral_amd import server as SV

    kw = dict(t_gsw=4)
    po, pg = O.make_params(5, 3, **kw), sa.make_params(5, 3, **kw)
    db = O.gen_db(po, 31)
    owner, lane = sa.Server(pg), sa.Server(pg)
    owner.gen_db(31)
    # the lane first answers from an image of its own through captured graphs: share_db must not leave them pointing at it
    warm = O.Client(po, seed=9)
    lane.gen_db(99)
    lane.set_pub_params(*warm.pub_params())
    lane.use_graphs(True)
    lane.set_query(warm.query(1))
    lane.run_query()
    lane.sync()
    lane.share_db(owner)
    with pytest.raises(RuntimeError):
        lane.gen_db(5)
    with pytest.raises(RuntimeError):
        lane.fill_db_random(5)
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    clients = [O.Client(po, seed=5), O.Client(po, seed=6)]
    pps = [cl.pub_params() for cl in clients]  # (fresh randomness per call: generated once)
    for s, st, pp in zip((owner, lane), streams, pps):
        s.set_stream(st.cuda_stream)
        s.set_pub_params(*pp)
        s.use_graphs(True)
    for rnd, (i0, i1) in enumerate([(0, 255), (17, 17), (200, 3), (128, 64)]):
        qs = [clients[0].query(i0), clients[1].query(i1)]
        owner.set_query(qs[0])
        lane.set_query(qs[1])
        for _ in range(3):  # several replays in flight on both streams before anything is read
            owner.run_query()
            lane.run_query()
        owner.sync()
        lane.sync()
        for s, cl, pp, q, idx in zip((owner, lane), clients, pps, qs, (i0, i1)):
            assert_eq(s.read(SV.BUF_FINAL), O.answer(po, q, *pp, db), f"lane answer round {rnd} idx={idx}")
            assert_eq(cl.decode(s.read(SV.BUF_RESPONSE)), O.db_item(po, 31, idx), "decoded plaintext")
    lane.close()
    fin, resp, _ = owner.answer(clients[0].query(9))  # the owner is unaffected by the lane's release
    assert_eq(clients[0].decode(resp), O.db_item(po, 31, 9), "owner after the lane closed")
    owner.close()


def test_lane_created_on_the_owners_image(sa, oracle):
    """create_lane (Server(share_db_of=...)): a query lane that never allocates an image of its own; a server with another
    plaintext modulus cannot share (its response switch would use the wrong modulus); column read-back == slot read-back"""
    O = oracle
    from spiral_amd import server as SV

    kw = dict(t_gsw=4)
    po, pg = O.make_params(5, 3, **kw), sa.make_params(5, 3, **kw)
    db = O.gen_db(po, 31)
    owner = sa.Server(pg)
    with pytest.raises(RuntimeError):
        sa.Server(pg, share_db_of=owner)  # nothing loaded yet
    owner.gen_db(31)
    lane = sa.Server(pg, share_db_of=owner)
    with pytest.raises(RuntimeError):
        lane.gen_db(5)
    with pytest.raises(RuntimeError):
        sa.Server(pg, share_db_of=lane)  # a lane does not own the image
    other = sa.Server(sa.make_params(5, 3, t_gsw=4, p_db=16))
    with pytest.raises(RuntimeError):
        other.share_db(owner)
    other.close()
    cl = O.Client(po, seed=8)
    pp = cl.pub_params()
    lane.set_pub_params(*pp)
    for idx in (0, 77, 255):
        q = cl.query(idx)
        fin, resp, _ = lane.answer(q)
        assert_eq(fin, O.answer(po, q, *pp, db), f"lane answer idx={idx}")
        assert_eq(cl.decode(resp), O.db_item(po, 31, idx), "decoded plaintext")
    slabs = owner.read_db_slots(0, N)  # [N][num_per][2][dim0][2]
    for ii0, n in ((0, 1), (7, 1), (2, 3)):
        assert_eq(owner.read_db_columns(ii0, n), slabs[:, ii0:ii0 + n], f"columns {ii0}..{ii0 + n - 1}")
    with pytest.raises(RuntimeError):
        owner.read_db_columns(7, 2)
    lane.close()
    owner.close()


@pytest.mark.parametrize("nu1,nu2,n", [(4, 5, 2), (3, 6, 3), (3, 5, 4), (5, 3, 3), (6, 6, 4), (6, 6, 7)])
def test_batched_sweep_equals_single_sweeps(sa, oracle, nu1, nu2, n):
    """first_dim_batch: the queries of n lanes against one pass over the shared database image; every lane's accumulators and
    final answer equal its own first_dim() and the oracle.  (5, 3): fewer than 64 output columns, where the call falls back to
    one sweep per lane; (6, 6): one pass on the matrix cores (sweep_mfma.hip), the others passes of two on the vector ALU.
    Lanes run on their own streams with the whole-group graphs either side of the shared sweep."""
    import torch

    O = oracle
    from spiral_amd import server as SV

    kw = dict(t_gsw=8 if nu2 >= 5 else 4)  # (t_gsw = 4 is bit-exact too but too noisy to decode from nu2 = 5 up)
    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    s = O.shape_of(po)
    db = O.gen_db(po, 77)
    owner = sa.Server(pg)
    owner.gen_db(77)
    lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(n - 1)]
    clients = [O.Client(po, seed=40 + b) for b in range(n)]
    pps = [cl.pub_params() for cl in clients]
    streams = [torch.cuda.Stream() for _ in range(n)]
    for srv, st, pp in zip(lanes, streams, pps):
        srv.set_stream(st.cuda_stream)
        srv.set_pub_params(*pp)
        srv.use_graphs(True)
    total = s.dim0 * s.num_per
    for rnd in range(2):
        idxs = [(17 * rnd + 5 * b) % total for b in range(n)]
        qs = [cl.query(i) for cl, i in zip(clients, idxs)]
        for srv, q in zip(lanes, qs):
            srv.set_query(q)
            srv.run_pre()
        sa.first_dim_batch(lanes)
        for srv in lanes:
            srv.run_post()
        for srv in lanes:
            srv.sync()
        for b, (srv, cl, pp, q, idx) in enumerate(zip(lanes, clients, pps, qs, idxs)):
            cv = O.stage_expand(po, q, pp[0], pp[1])
            cts, gsw = O.stage_convert(po, cv, pp[2], pp[3])
            want_acc = O.multiply_query_by_database(O.reorient_ciphertexts(cts), db, s.dim0, s.num_per)
            assert_eq(srv.read(SV.BUF_ACC), want_acc, f"round {rnd} lane {b}: accumulators of the batched sweep")
            assert_eq(srv.read(SV.BUF_FINAL), O.stage_fold(po, O.from_ntt(want_acc), gsw), f"round {rnd} lane {b}: final ciphertext")
            assert_eq(cl.decode(srv.read(SV.BUF_RESPONSE)), O.db_item(po, 77, idx), f"round {rnd} lane {b}: decoded plaintext")
    with pytest.raises(RuntimeError):
        sa.first_dim_batch([lanes[0], lanes[0]])
    other = sa.Server(pg)
    other.gen_db(77)  # same contents, another image
    with pytest.raises(RuntimeError):
        sa.first_dim_batch([lanes[0], other])
    other.close()
    # every lane is validated before anything is launched: a lane whose new query has not been converted fails the call and no
    # lane is swept (the accumulators stay what the last round left)
    before = [srv.read(SV.BUF_ACC).copy() for srv in lanes]
    lanes[-1].set_query(clients[-1].query(1))
    for srv in lanes[:-1]:
        srv.run_pre()
    with pytest.raises(RuntimeError, match="has not converted"):
        sa.first_dim_batch(lanes)
    for srv, acc0 in zip(lanes, before):
        srv.sync()
        assert_eq(srv.read(SV.BUF_ACC), acc0, "accumulators untouched by the refused batch")
    # lifetime: closing the owner first keeps the image alive for its lanes (freed with the last of them)
    owner.close()
    if n > 1:
        lane, cl, pp = lanes[1], clients[1], pps[1]
        q = cl.query(3)
        lane.set_query(q)
        lane.run_query()
        lane.sync()
        assert_eq(lane.read(SV.BUF_FINAL), O.answer(po, q, pp[0], pp[1], pp[2], pp[3], db), "a lane answers after its owner was closed")
    for srv in lanes[1:]:
        srv.close()


_N_FUZZ_SWEEP = max(6, _N_FUZZ // 3)


@pytest.mark.parametrize("seed", range(_N_FUZZ_SWEEP))
def test_matrix_core_sweep_random_geometries_and_edge_values(sa, oracle, seed):
    """seeded random (geometry, queries per pass) for the matrix-core sweep with the limb decomposition's edge residues sprinkled over uniform ones in
    both operands: 0, 1, m - 1, the residues either side of the wrap point 2^28 - 0x808080, all-(-128) / all-(+127) limb bytes.  Against the oracle."""
    O = oracle
    rng = np.random.default_rng(7000 + seed)
    nu1, nu2 = int(rng.integers(6, 8)), int(rng.integers(6, 8))
    n = int(rng.integers(1, 9))
    dim0, num_per = 1 << nu1, 1 << nu2
    wrap = (1 << 28) - 0x808080
    edges = {0: [0, 1, O.P - 1, O.P - 2, wrap - 1, wrap, wrap + 1, (15 << 24) - 0x808080, (14 << 24) - 0x808080 + 0xFFFFFF, 0x808080, 0x7F7F7F],
             1: [0, 1, O.B - 1, O.B - 2, (14 << 24) - 0x808080, (13 << 24) - 0x808080 + 0xFFFFFF, 0x808080, 0x7F7F7F, 0x800000, 0x7FFFFF, 1 << 27]}

    def sprinkle(a, limb_axis):  # a[..., limb, z]: replace ~3 % of the residues by edge values of their prime
        for limb, mod in ((0, O.P), (1, O.B)):
            view = np.moveaxis(a, limb_axis, 0)[limb]
            mask = rng.random(view.shape) < 0.03
            vals = np.array([v for v in edges[limb] if v < mod], dtype=np.uint64)
            view[mask] = vals[rng.integers(0, len(vals), size=int(mask.sum()))]

    res = []
    for _ in range(n):
        cts = rand_ntt(rng, O, (dim0, 3, 2))
        sprinkle(cts, -2)
        res.append(O.reorient_ciphertexts(cts))
    lo = rng.integers(0, O.P, size=dim0 * num_per * 4 * N, dtype=np.uint64)
    hi = rng.integers(0, O.B, size=dim0 * num_per * 4 * N, dtype=np.uint64)
    both = np.stack([lo, hi])
    sprinkle(both[:, None, :], 0)
    db = both[0] | (both[1] << np.uint64(32))
    got = sa.multiplyQueriesByDatabase(res, db, dim0, num_per)
    for b, re in enumerate(res):
        assert_eq(got[b], O.multiply_query_by_database(re, db, dim0, num_per), f"seed {seed}: ({nu1},{nu2}), query {b} of {n}")


@pytest.mark.parametrize("G", [2, 8])
def test_matrix_core_sweep_groups_accumulators_by_fold_rank(sa, oracle, G):
    """first_dim_batch on lanes prepared for a distributed fold over G ranks (set_fold_ranks: the sweep writes ciphertext ii = g + G k at [g][k], sweep.hip
    acc_pos): the matrix-core sweep honours the same layout -- every lane's accumulator words equal what its own first_dim() writes."""
    import torch

    pg = sa.make_params(6, 6, t_gsw=8)
    s = sa.get_shape(pg)
    rng = np.random.default_rng(77)
    mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
    owner = sa.Server(pg)
    owner.fill_db_random(5)
    lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(2)]
    accs, want = [], []
    for ln in lanes:
        ln.set_fold_ranks(G)
        accs.append(torch.zeros(s.num_per * 6 * N, dtype=torch.int64, device="cuda"))
        ln.set_acc(accs[-1].data_ptr())
        ln.set_pub_params(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 2 * pg.t_conv)), mk((3, 2 * pg.t_conv)))
        ln.set_query(mk((s.n_query_cts, 2)))
        ln.run_pre()
        ln.first_dim()
        ln.sync()
        torch.cuda.synchronize()
        want.append(accs[-1].clone())
        accs[-1].fill_(-1)
    torch.cuda.synchronize()
    sa.first_dim_batch(lanes)
    for ln in lanes:
        ln.sync()
    torch.cuda.synchronize()
    for b, (got, w) in enumerate(zip(accs, want)):
        assert torch.equal(got, w), f"lane {b}: accumulators of the batched sweep differ from first_dim() with {G} fold ranks"
    assert not torch.equal(want[0], want[1])
    for ln in reversed(lanes):
        ln.close()


def test_first_dim_batch_on_a_shard_that_is_not_a_power_of_two(sa, oracle):
    """A server may hold any first-dimension range [j0, j1).  192 of 256 indices is a multiple of 64 but not a power of two: the matrix-core kernel walks
    a work item's pieces with shifts and masks and must NOT take such a shard (sweep_mfma_ok) -- the batch falls back to vector-ALU passes and every
    lane's accumulators are the oracle's partial sums over j in [0, 192).  The complementary shard [192, 256) (64 indices: a power of two) does take it."""
    O = oracle
    from spiral_amd import server as SV

    nu1, nu2, n = 8, 6, 3
    pg = sa.make_params(nu1, nu2, t_gsw=8)
    s = sa.get_shape(pg)
    rng = np.random.default_rng(4242)
    db = O.fill_db_random(77, s.dim0 * s.num_per * 4 * N).reshape(N, s.num_per, 2, s.dim0, 2)
    mk = lambda shape: np.stack([rng.integers(0, m, size=shape + (N,), dtype=np.uint64) for m in (sa.P, sa.B)], axis=-2)
    pps = [(mk((s.n_left, 2, pg.t_exp)), mk((s.n_right, 2, pg.t_exp_right)), mk((3, 2 * pg.t_conv)), mk((3, 2 * pg.t_conv))) for _ in range(n)]
    qs = [mk((s.n_query_cts, 2)) for _ in range(n)]
    po = O.make_params(nu1, nu2, t_gsw=8)
    for j0, j1, fmt in ((0, 192, SV.DB_PACKED), (192, 256, SV.DB_LIMBS)):
        owner = sa.Server(pg, j_begin=j0, j_end=j1)
        owner.load_db(db)
        lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(n - 1)]
        for ln, pp, q in zip(lanes, pps, qs):
            ln.keep_cts(True)
            ln.set_pub_params(*pp)
            ln.set_query(q)
            ln.run_pre()
        sa.first_dim_batch(lanes)
        assert owner.db_format() == fmt, f"shard [{j0}, {j1})"
        sub = np.ascontiguousarray(db[:, :, :, j0:j1, :])
        for b, ln in enumerate(lanes):
            ln.sync()
            cts = ln.read(SV.BUF_CTS)  # this shard's first-dimension ciphertexts, as the sweep consumed them
            want = O.multiply_query_by_database(O.reorient_ciphertexts(cts), sub, j1 - j0, s.num_per)
            assert_eq(ln.read(SV.BUF_ACC), want, f"shard [{j0}, {j1}), lane {b}")
        for ln in reversed(lanes):
            ln.close()


def test_db_format_round_trip(sa, oracle, opts):
    """The one database image in its two forms (include/spiral_gpu.h spiral_gpu_server_set_db_format).  packed -> limb planes -> packed in place, through a
    staging buffer smaller than the image; in either form: read_db_slots / read_db_item give the oracle's database, a single query gives the oracle's answer
    (vector-ALU kernel on the packed form, the one-query matrix-core kernel on the limb planes), captured graphs follow the form; a partial load_db_items and
    set_sweep_stages take the image back to the packed form by themselves; device bytes held = ONE image throughout."""
    O = oracle
    from spiral_amd import server as SV

    kw = dict(t_gsw=8)
    po, pg = O.make_params(6, 6, **kw), sa.make_params(6, 6, **kw)
    s = O.shape_of(po)
    total = s.dim0 * s.num_per
    db = O.gen_db(po, 21)
    db5 = db.reshape(N, s.num_per, 2, s.dim0, 2)
    cl = O.Client(po, seed=9)
    pp = cl.pub_params()
    srv = sa.Server(pg)
    srv.gen_db(21)
    srv.set_pub_params(*pp)
    srv.use_graphs(True)
    image = srv.db_device_bytes()
    assert image == N * s.dim0 * s.num_per * 4 * 7, "3.5 bytes per residue"
    lane = sa.Server(pg, share_db_of=srv)
    lane.set_pub_params(*pp)
    lane.use_graphs(True)
    with pytest.raises(RuntimeError, match="owner"):
        lane.set_db_format(SV.DB_LIMBS)

    def check(tag, idx):
        q = cl.query(idx)
        want = O.answer(po, q, *pp, db)
        for who, sv in (("owner", srv), ("lane", lane)):
            sv.set_query(q)
            for rnd in range(2):  # capture (or re-capture after a change of form), replay
                sv.run_query()
                sv.sync()
                assert_eq(sv.read(SV.BUF_FINAL), want, f"{tag}: {who}, run {rnd}")
        assert_eq(cl.decode(srv.read(SV.BUF_RESPONSE)), O.db_item(po, 21, idx), f"{tag}: decoded item")
        assert_eq(srv.read_db_slots(5, 3), db5[5:8], f"{tag}: read_db_slots")
        assert_eq(lane.read_db_item(idx), O.encode_item(po, O.db_item(po, 21, idx)), f"{tag}: read_db_item")
        assert srv.db_device_bytes() == image == lane.db_device_bytes(), f"{tag}: one image"

    check("packed", 1234)
    for rnd in range(2):
        srv.set_db_format(SV.DB_LIMBS)
        assert lane.db_format() == SV.DB_LIMBS
        check(f"limb planes {rnd}", 77 + rnd)
        srv.set_db_format(SV.DB_PACKED)
        check(f"packed again {rnd}", total - 1 - rnd)
    # a batch converts by itself; a staged sweep and a partial reload convert back
    srv.set_query(cl.query(5))
    lane.set_query(cl.query(6))
    sa.run_query_batch([srv, lane])
    assert srv.db_format() == SV.DB_LIMBS
    check("after a batch", 99)
    srv.set_fold_ranks(2)
    srv.set_sweep_stages(2)
    assert srv.db_format() == SV.DB_PACKED, "stage-by-stage launches read the packed form"
    srv.set_fold_ranks(1)
    check("after set_sweep_stages", 100)
    srv.set_db_format(SV.DB_LIMBS)
    items = np.stack([O.db_item(po, 22, i) for i in range(total // 2, total // 2 + s.num_per)])  # one first-dimension index of ANOTHER database
    srv.load_db_items(O.pack_items(items, 8), 8, first_item=total // 2)
    assert srv.db_format() == SV.DB_PACKED, "a partial load scatters packed words"
    j = (total // 2) // s.num_per
    got = srv.read_db_slots(0, N)
    want_db = db5.copy()
    for i in range(s.num_per):
        enc = O.encode_item(po, items[i])  # [m][c][limb][z]
        want_db[:, i, :, j, :] = (enc[:, :, 0, :] | (enc[:, :, 1, :] << np.uint64(32))).transpose(2, 1, 0)
    assert_eq(got, want_db, "the image after a partial load into a converted image")
    lane.close()
    srv.close()


@pytest.mark.parametrize("nu1,nu2,kw,graphs", [
    (3, 2, dict(t_gsw=5, t_exp=2, qprime_bits=19, direct_upload=1), True),   # the SpiralStream form (configs[3]): no expansion
    (4, 3, dict(t_gsw=8), True),                                             # query compression: expansion + conversion run ONCE per rank
    (6, 6, dict(t_gsw=8), False),                                            # a geometry with both image forms: instance 1's image in limb planes, the others packed
])
def test_run_query_instances_factor_3_over_two_ranks(sa, oracle, nu1, nu2, kw, graphs):
    """configs[3] whole: an item larger than one plaintext is factor = ceil(item / plaintext) database instances (select_params.py:297-298, 409-418).  Here
    factor 3 over 2 emulated ranks on one device: rank r holds instances r, r + 2 (own images) and one query server; every rank converts the ONE query
    once and answers it against its instances (spiral_gpu_server_run_query_instances); the responses, gathered as spiral_amd/dist.py lays them out, decode
    through the client to the three plaintexts of the item.  Folded ciphertexts and responses against the oracle's answer per instance."""
    import torch

    O = oracle
    from spiral_amd import dist as sdist
    from spiral_amd import server as SV

    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    s = O.shape_of(po)
    factor, world = 3, 2
    slots = (factor + world - 1) // world
    cl = O.Client(po, seed=12)
    pp = cl.pub_params()
    words = 6 * N
    blocks = []
    idxs = [5, (s.dim0 * s.num_per) - 1]
    ranks = []
    for r in range(world):
        inst = []
        for k in sdist.instances_of_rank(r, world, factor):
            sv = sa.Server(pg)
            sv.gen_db(100 + k)
            if k == 1 and (nu1, nu2) == (6, 6):
                sv.set_db_format(SV.DB_LIMBS)
            inst.append(sv)
        qsrv = sa.Server(pg, share_db_of=inst[0])  # the query's server: a lane of this rank's first instance (no image of its own)
        qsrv.set_pub_params(*pp)
        qsrv.use_graphs(graphs)
        ranks.append((qsrv, inst, torch.zeros(slots * words, dtype=torch.int64, device="cuda"), torch.zeros(slots * words, dtype=torch.int64, device="cuda")))
    for rnd, idx in enumerate(idxs * 2):  # second pass: graph replays with new queries
        q = cl.query(idx)
        for qsrv, inst, resp, fins in ranks:
            qsrv.set_query(q)
            if rnd % 2 == 0:
                qsrv.run_query_instances(inst, resp.data_ptr(), fins.data_ptr(), pre=True)
            else:  # conversion as its own step, then the instances
                qsrv.run_pre()
                qsrv.run_query_instances(inst, resp.data_ptr(), fins.data_ptr(), pre=False)
            qsrv.sync()
        gathered = torch.cat([r[2] for r in ranks])  # what all_gather_instance_responses produces: [rank][slot][6 x 2048]
        gathered_f = torch.cat([r[3] for r in ranks])
        for k in range(factor):
            want = O.answer(po, q, *pp, O.gen_db(po, 100 + k))
            fin = sdist.instance_response(gathered_f, k, world, slots).cpu().numpy().view(np.uint64).reshape(3, 2, N)
            resp = sdist.instance_response(gathered, k, world, slots).cpu().numpy().view(np.uint64).reshape(3, 2, N)
            assert_eq(fin, want, f"round {rnd}, instance {k}: folded ciphertext")
            assert_eq(resp, O.stage_rescale(po, want), f"round {rnd}, instance {k}: response")
            assert_eq(cl.decode(resp), O.db_item(po, 100 + k, idx), f"round {rnd}, instance {k}: plaintext {k} of item {idx}")
    # an instance of another geometry or without a database is refused before anything is launched
    other = sa.Server(sa.make_params(nu1, nu2 + 1, **kw))
    other.gen_db(1)
    with pytest.raises(RuntimeError, match="differs"):
        ranks[0][0].run_query_instances([ranks[0][1][0], other], ranks[0][2].data_ptr())
    other.close()
    for qsrv, inst, _, _ in ranks:
        qsrv.close()
        for sv in inst:
            sv.close()


_N_FUZZ_INST = max(4, _N_FUZZ // 8)


@pytest.mark.parametrize("nu1,nu2,kw", _with_matrix_core_geometries(_random_parameter_sets(_N_FUZZ_INST, 7117)), ids=[f"set{i}" for i in range(_N_FUZZ_INST)])
def test_random_parameter_sets_instances(sa, oracle, nu1, nu2, kw):
    """a third seeded draw through spiral_gpu_server_run_query_instances: one query against two database instances (factor 2; one of them in the limb-plane form
    where the geometry has one), eagerly and as a replayed hipGraph, conversion inside the call and as its own step: every instance's folded ciphertext and
    response == the oracle's answer against that instance's database"""
    import torch

    O = oracle
    from spiral_amd import server as SV

    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    cl = O.Client(po, seed=3 + nu1 + 5 * nu2)
    pp = cl.pub_params()
    inst, dbs = [], []
    for k in range(2):
        sv = sa.Server(pg)
        sv.gen_db(40 + k)
        dbs.append(O.gen_db(po, 40 + k))
        inst.append(sv)
    if _has_limb_form(nu1, nu2):  # instance 1 answers from the limb planes through sweep_mfma_kernel<1>
        inst[1].set_db_format(SV.DB_LIMBS)
        assert inst[1].db_format() == SV.DB_LIMBS
    inst[0].set_pub_params(*pp)  # the query lives on instance 0's server
    words = 6 * N
    resp = torch.zeros(2 * words, dtype=torch.int64, device="cuda")
    fins = torch.zeros(2 * words, dtype=torch.int64, device="cuda")
    total = 1 << (nu1 + nu2)
    for rnd, idx in enumerate((1 % total, total - 1, total // 2, 0)):
        q = cl.query(idx)
        inst[0].use_graphs(rnd >= 1)
        inst[0].set_query(q)
        if rnd == 2:
            inst[0].run_pre()
        inst[0].run_query_instances(inst, resp.data_ptr(), fins.data_ptr(), pre=rnd != 2)
        inst[0].sync()
        for k in range(2):
            want = O.answer(po, q, *pp, dbs[k])
            got_f = fins[k * words:(k + 1) * words].cpu().numpy().view(np.uint64).reshape(3, 2, N)
            got_r = resp[k * words:(k + 1) * words].cpu().numpy().view(np.uint64).reshape(3, 2, N)
            assert_eq(got_f, want, f"round {rnd} instance {k}: folded ciphertext, params {nu1},{nu2},{kw}")
            assert_eq(got_r, O.stage_rescale(po, want), f"round {rnd} instance {k}: response, params {nu1},{nu2},{kw}")
    for sv in inst:
        sv.close()


def test_batch_without_the_matrix_core_image_and_after_a_reload(sa, oracle, opts):
    """(6, 6) is a geometry the matrix-core sweep covers.  Option sweep_mfma_min = 0 (taken when a handle is created): the same batch sweeps in passes of two on
    the vector ALU (2 + 2 + 1 for five lanes).  With the image: reloading the database drops it -- the next batch rebuilds it from the new contents."""
    O = oracle
    from spiral_amd import server as SV

    kw = dict(t_gsw=8)
    po, pg = O.make_params(6, 6, **kw), sa.make_params(6, 6, **kw)
    s = O.shape_of(po)
    n = 5
    clients = [O.Client(po, seed=300 + b) for b in range(n)]
    pps = [cl.pub_params() for cl in clients]
    qs = [cl.query((97 * b + 5) % (s.dim0 * s.num_per)) for b, cl in enumerate(clients)]

    def answers(seed):
        db = O.gen_db(po, seed)
        return [O.answer(po, q, *pp, db) for q, pp in zip(qs, pps)]

    def lanes_on(seed):
        owner = sa.Server(pg)
        owner.gen_db(seed)
        lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(n - 1)]
        for srv, pp, q in zip(lanes, pps, qs):
            srv.set_pub_params(*pp)
            srv.set_query(q)
            srv.use_graphs(True)
        return lanes

    want = {seed: answers(seed) for seed in (11, 12)}
    opts(sweep_mfma_min=0)
    lanes = lanes_on(11)
    sa.run_query_batch(lanes)
    for b, srv in enumerate(lanes):
        srv.sync()
        assert_eq(srv.read(SV.BUF_FINAL), want[11][b], f"vector-ALU passes, lane {b}")
    assert lanes[0].db_format() == SV.DB_PACKED, "no matrix-core sweep, no conversion"
    for srv in reversed(lanes):
        srv.close()
    for one_image in (1, 0):  # the image converted in place (default) / a second image beside the packed one
        opts(sweep_mfma_min=2, one_image=one_image)
        lanes = lanes_on(11)
        image = lanes[0].db_device_bytes()
        for seed in (11, 12, 11):  # the limb planes follow the database: built, rebuilt after each reload
            lanes[0].gen_db(seed)
            assert lanes[-1].db_format() == SV.DB_PACKED, "a loader leaves the packed form"
            for rnd in range(2):  # capture, replay
                sa.run_query_batch(lanes)
                for b, srv in enumerate(lanes):
                    srv.sync()
                    assert_eq(srv.read(SV.BUF_FINAL), want[seed][b], f"one_image {one_image}, database {seed}, run {rnd}, lane {b}")
            assert lanes[-1].db_format() == (SV.DB_LIMBS if one_image else SV.DB_PACKED)
            assert lanes[-1].db_device_bytes() == (image if one_image else 2 * image), "device bytes held for database images"
        for srv in reversed(lanes):
            srv.close()


@pytest.mark.parametrize("nu1,nu2,n,kw,graphs", [
    (3, 6, 2, dict(t_gsw=8), True), (3, 6, 4, dict(t_gsw=8), True), (5, 6, 2, dict(t_gsw=8), False), (5, 6, 4, dict(t_gsw=8), True),
    (5, 3, 3, dict(t_gsw=4), True),                                                       # fewer than 64 output columns: one sweep per lane inside the sequence
    (2, 2, 4, dict(t_gsw=5, t_exp=2, qprime_bits=19, direct_upload=1), True),             # no expansion (SpiralStream form), plain database layout
    (4, 5, 3, dict(t_gsw=14), True),                                                      # a gadget whose digits do not recompose: the two-product fold (fold_chain_kernel) with lanes
    (2, 1, 2, {}, False),                                                                  # stopround == 0
    (3, 6, 8, dict(t_gsw=8), True), (5, 6, 5, dict(t_gsw=8), True), (4, 5, 7, dict(t_gsw=8), False),  # more than two lanes: the vector-ALU sweep in passes of two
    (6, 6, 3, dict(t_gsw=8), True), (6, 6, 8, dict(t_gsw=8), True), (7, 6, 5, dict(t_gsw=8), False),  # the sweep on the matrix cores, one pass for all lanes
    (5, 3, 6, dict(t_gsw=4), True),                                                       # six lanes on the per-lane sweep fallback
])
def test_run_query_batch_equals_single_queries(sa, oracle, nu1, nu2, n, kw, graphs):
    """run_query_batch: n whole queries (different clients: own keys, own query) in one launch sequence whose every launch carries all of them
    (gridDim.z = n; the reference answers one per process_crtd_query, src/spiral.cpp:2337-2406).  Every lane's expanded ciphertexts, GSW
    matrices, accumulators, final ciphertext and response equal the oracle's for ITS inputs, and the lane's own run_query reproduces them."""
    O = oracle
    from spiral_amd import server as SV

    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    s = O.shape_of(po)
    db = O.gen_db(po, 77)
    owner = sa.Server(pg)
    owner.gen_db(77)
    lanes = [owner] + [sa.Server(pg, share_db_of=owner) for _ in range(n - 1)]
    clients = [O.Client(po, seed=70 + b) for b in range(n)]
    pps = [cl.pub_params() for cl in clients]
    for srv, pp in zip(lanes, pps):
        srv.set_pub_params(*pp)
        srv.use_graphs(graphs)
    total = s.dim0 * s.num_per
    decodes = kw.get("t_gsw", 8) >= 8 or nu2 <= 3  # (small gadgets are bit-exact too but too noisy to decode at large nu2)
    for rnd in range(3):  # round 0 captures the graph, 1 and 2 replay it on new queries
        idxs = [(29 * rnd + 11 * b + 3) % total for b in range(n)]
        qs = [cl.query(i) for cl, i in zip(clients, idxs)]
        for srv, q in zip(lanes, qs):
            srv.set_query(q)
        sa.run_query_batch(lanes)
        for srv in lanes:
            srv.sync()
        for b, (srv, cl, pp, q, idx) in enumerate(zip(lanes, clients, pps, qs, idxs)):
            what = f"round {rnd} lane {b} of {n}"
            cv = O.stage_expand(po, q, pp[0], pp[1])
            assert_eq(srv.read(SV.BUF_EXPANDED), cv, f"{what}: expanded ciphertexts")
            cts, gsw = O.stage_convert(po, cv, pp[2], pp[3])
            assert_eq(srv.read(SV.BUF_GSW), gsw, f"{what}: regevToGSW outputs")
            want_acc = O.multiply_query_by_database(O.reorient_ciphertexts(cts), db, s.dim0, s.num_per)
            assert_eq(srv.read(SV.BUF_ACC), want_acc, f"{what}: accumulators")
            fin = O.stage_fold(po, O.from_ntt(want_acc), gsw)
            assert_eq(srv.read(SV.BUF_FINAL), fin, f"{what}: final ciphertext")
            resp = srv.read(SV.BUF_RESPONSE)
            assert_eq(resp, O.stage_rescale(po, fin), f"{what}: response")
            if decodes:
                assert_eq(cl.decode(resp), O.db_item(po, 77, idx), f"{what}: decoded plaintext")
    # the lanes stay ordinary servers: a lane's own run_query after a batch gives the same answer again
    last = [(srv.read(SV.BUF_FINAL).copy(), srv.read(SV.BUF_RESPONSE).copy()) for srv in lanes]
    for srv, (fin, resp) in zip(lanes, last):
        srv.run_query()
        srv.sync()
        assert_eq(srv.read(SV.BUF_FINAL), fin, "single run_query after the batch: final ciphertext")
        assert_eq(srv.read(SV.BUF_RESPONSE), resp, "single run_query after the batch: response")
    # a different lane set re-captures; a sub-batch answers the same
    if n >= 3:
        sa.run_query_batch(lanes[1:])
        for srv, (fin, resp) in zip(lanes[1:], last[1:]):
            srv.sync()
            assert_eq(srv.read(SV.BUF_FINAL), fin, "sub-batch led by another lane: final ciphertext")
    with pytest.raises(RuntimeError, match="listed twice"):
        sa.run_query_batch([lanes[0], lanes[0]])
    other = sa.Server(pg)
    other.gen_db(77)  # same contents, another image
    other.set_pub_params(*pps[0])
    other.set_query(clients[0].query(0))
    with pytest.raises(RuntimeError, match="database image"):
        sa.run_query_batch([lanes[0], other])
    other.close()
    lanes[-1].keep_cts(True)
    with pytest.raises(RuntimeError, match="keep_cts"):
        sa.run_query_batch(lanes)
    for srv in lanes[1:]:
        srv.close()
    owner.close()


def test_sharded_first_dim_sums_to_unsharded(sa, oracle):
    """two j-shards on one device: summing their accumulators (what the RCCL reduce does) == one server"""
    O = oracle
    from spiral_amd import server as SV

    kw = dict(t_gsw=4)
    po, pg = O.make_params(4, 3, **kw), sa.make_params(4, 3, **kw)
    cl = O.Client(po, seed=2)
    wl, wr, w, v = cl.pub_params()
    q = cl.query(99)
    accs = []
    for j0, j1 in [(0, 16), (0, 8), (8, 16)]:
        srv = sa.Server(pg, 0, j0, j1)
        srv.gen_db(55)
        srv.set_pub_params(wl, wr, w, v)
        srv.set_query(q)
        srv.expand()
        srv.convert()
        srv.first_dim()
        accs.append(srv.read(SV.BUF_ACC).astype(object))
        srv.close()
    tot = accs[1] + accs[2]
    tot[..., 0, :] %= O.P
    tot[..., 1, :] %= O.B
    assert (tot == accs[0]).all()


@pytest.mark.parametrize("G,graphs", [(2, False), (4, False), (8, False), (4, True)])
def test_distributed_fold_emulated_on_one_gpu(sa, oracle, G, graphs):
    """j-shards + reduce-scatter + local folds + gather + root folds, the ranks emulated by G servers on one device
    and the two collectives by torch sums/slices: must equal the single-device answer bit for bit.  With graphs on,
    run_pre_sweep / fold_local / fold_root replay as hipGraphs across several queries (fixed buffers, as bench.py)."""
    import torch
    from spiral_amd import server as SV

    O = oracle
    kw = dict(t_gsw=4)
    po, pg = O.make_params(4, 4, **kw), sa.make_params(4, 4, **kw)
    s = O.shape_of(po)
    cl = O.Client(po, seed=31)
    wl, wr, w, v = cl.pub_params()
    db = O.gen_db(po, 77)
    dev = torch.device("cuda", 0)
    words = s.num_per * 6 * N
    L = words // G
    srvs, accs, chunks, cts = [], [], [], []
    for g in range(G):
        srv = sa.Server(pg, 0, g * s.dim0 // G, (g + 1) * s.dim0 // G)
        srv.gen_db(77)
        srv.set_pub_params(wl, wr, w, v)
        srv.set_fold_ranks(G)
        accs.append(torch.zeros(words, dtype=torch.int64, device=dev))
        chunks.append(torch.zeros(L, dtype=torch.int64, device=dev))
        cts.append(torch.zeros(6 * N, dtype=torch.int64, device=dev))
        srv.set_acc(accs[g].data_ptr())
        srv.use_graphs(graphs)
        srvs.append(srv)
    gathered = torch.zeros(G * 6 * N, dtype=torch.int64, device=dev)
    for idx in ((201, 7, 255) if graphs else (201,)):
        q = cl.query(idx)
        for g in range(G):
            srvs[g].set_query(q)
            if graphs:
                srvs[g].run_pre_sweep()
            else:
                srvs[g].run_pre()
                srvs[g].first_dim()
            srvs[g].sync()
        total = torch.stack(accs).sum(0)  # the reduce part
        for g in range(G):  # the scatter part + local folds
            chunks[g].copy_(total[g * L:(g + 1) * L])
            torch.cuda.synchronize()
            srvs[g].fold_local(chunks[g].data_ptr(), cts[g].data_ptr())
            srvs[g].sync()
        gathered.copy_(torch.cat(cts))  # the all-gather
        torch.cuda.synchronize()
        srvs[0].fold_root(gathered.data_ptr())
        srvs[0].sync()
        assert_eq(srvs[0].read(SV.BUF_FINAL), O.answer(po, q, wl, wr, w, v, db), f"distributed fold G={G} idx={idx}")
        assert_eq(cl.decode(srvs[0].read(SV.BUF_RESPONSE)), O.db_item(po, 77, idx), "decoded plaintext")
    for srv in srvs:
        srv.close()


_N_FUZZ_SHARD = max(6, _N_FUZZ // 8)


def _shardable(sets):
    return [(nu1, max(nu2, 1), kw) for nu1, nu2, kw in sets]  # (a fold over ranks needs at least two ciphertexts)


@pytest.mark.parametrize("nu1,nu2,kw", _shardable(_random_parameter_sets(_N_FUZZ_SHARD, 909)), ids=[f"set{i}" for i in range(_N_FUZZ_SHARD)])
def test_random_parameter_sets_sharded(sa, oracle, nu1, nu2, kw):
    """the N > 1 answer path on a seeded draw of parameter sets: G emulated ranks (2 or 4, as the geometry allows) with j-shards of the database, the reduce-scatter and
    the all-gather as torch sums / slices, local folds and the root fold as replayed hipGraphs (the call sequence of bench.py's in-order schedule) == the oracle's
    single-device answer, bit for bit -- odd gadget dimensions, both query forms, shards of one first-dimension index"""
    import torch
    from spiral_amd import dist as sdist
    from spiral_amd import server as SV

    O = oracle
    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    s = O.shape_of(po)
    G = 4 if (nu1 >= 2 and nu2 >= 2 and (nu1 + nu2) % 2 == 0) else 2
    cl = O.Client(po, seed=17 + nu1)
    wl, wr, w, v = cl.pub_params()
    db = O.gen_db(po, 55)
    dev = torch.device("cuda", 0)
    words = s.num_per * 6 * N
    L = words // G
    srvs, accs, chunks, cts = [], [], [], []
    for g in range(G):
        srv = sa.Server(pg, 0, g * s.dim0 // G, (g + 1) * s.dim0 // G)
        srv.gen_db(55)
        srv.set_pub_params(wl, wr, w, v)
        srv.set_fold_ranks(G)
        accs.append(torch.zeros(words, dtype=torch.int64, device=dev))
        chunks.append(torch.zeros(L, dtype=torch.int64, device=dev))
        cts.append(torch.zeros(6 * N, dtype=torch.int64, device=dev))
        srv.set_acc(accs[g].data_ptr())
        srv.use_graphs(True)
        srvs.append(srv)
    gathered = torch.zeros(G * 6 * N, dtype=torch.int64, device=dev)
    # where the query form allows it the expansion is sharded too (every rank its own subtree and every G-th GSW bit, the bits all-gathered): bench.py's default at N > 1
    shard_expand = sdist.expand_shard_ok(sa.get_shape(pg), pg, G)
    if shard_expand:
        for g in range(G):
            srvs[g].set_expand_shard(g, G)
        bits = [torch.zeros(srvs[g].gsw_bits_words(), dtype=torch.int64, device=dev) for g in range(G)]
        bits_all = torch.zeros(G * bits[0].numel(), dtype=torch.int64, device=dev)
    total_items = 1 << (nu1 + nu2)
    for idx in (total_items - 1, 0, total_items // 3):
        q = cl.query(idx)
        for g in range(G):
            srvs[g].set_query(q)
            if shard_expand:
                srvs[g].run_expand_pack(bits[g].data_ptr())
            else:
                srvs[g].run_pre_sweep()
            srvs[g].sync()
        if shard_expand:
            bits_all.copy_(torch.cat(bits))  # the all-gather of the GSW bits
            torch.cuda.synchronize()
            for g in range(G):
                srvs[g].run_unpack_convert_sweep(bits_all.data_ptr())
                srvs[g].sync()
        total = torch.stack(accs).sum(0)
        for g in range(G):
            chunks[g].copy_(total[g * L:(g + 1) * L])
            torch.cuda.synchronize()
            srvs[g].fold_local(chunks[g].data_ptr(), cts[g].data_ptr())
            srvs[g].sync()
        gathered.copy_(torch.cat(cts))
        torch.cuda.synchronize()
        srvs[0].fold_root(gathered.data_ptr())
        srvs[0].sync()
        want = O.answer(po, q, wl, wr, w, v, db)
        assert_eq(srvs[0].read(SV.BUF_FINAL), want, f"sharded over G={G} idx={idx}: folded ciphertext, params {nu1},{nu2},{kw}")
        assert_eq(srvs[0].read(SV.BUF_RESPONSE), O.stage_rescale(po, want), f"sharded over G={G} idx={idx}: response")
    for srv in srvs:
        srv.close()


@pytest.mark.parametrize("G,K,nu2", [(1, 2, 6), (2, 2, 6), (8, 2, 6), (2, 4, 7), (4, 4, 7)])
def test_pipelined_sweep_stages_emulated_on_one_gpu(sa, oracle, G, K, nu2):
    """the pipelined sweep: G emulated ranks sweep their j-shards in K column-block stages (first_dim_stage) into accumulators laid out
    [stage][rank][ct]; every stage's contiguous 1/K of the buffers is summed and scattered on its own (the per-stage reduce-scatter) into
    rows [s L/K, (s+1) L/K) of each rank's chunk; local folds, gather, root fold == the oracle's answer.  Also: first_dim() with stages
    set writes the same buffer as the K stage launches."""
    import torch
    from spiral_amd import server as SV

    O = oracle
    kw = dict(t_gsw=8)
    po, pg = O.make_params(6, nu2, **kw), sa.make_params(6, nu2, **kw)  # shards of dim0 / G >= 8: the packed database layout
    s = O.shape_of(po)
    cl = O.Client(po, seed=41)
    wl, wr, w, v = cl.pub_params()
    db = O.gen_db(po, 78)
    dev = torch.device("cuda", 0)
    words = s.num_per * 6 * N
    L = words // G
    srvs, accs, chunks, cts = [], [], [], []
    for g in range(G):
        srv = sa.Server(pg, 0, g * s.dim0 // G, (g + 1) * s.dim0 // G)
        srv.gen_db(78)
        srv.set_pub_params(wl, wr, w, v)
        srv.set_fold_ranks(G)
        assert srv.max_sweep_stages() == s.num_per // 32
        srv.set_sweep_stages(K)
        accs.append(torch.zeros(words, dtype=torch.int64, device=dev))
        chunks.append(torch.zeros(L, dtype=torch.int64, device=dev))
        cts.append(torch.zeros(6 * N, dtype=torch.int64, device=dev))
        srv.set_acc(accs[g].data_ptr())
        srv.use_graphs(True)
        srvs.append(srv)
    gathered = torch.zeros(G * 6 * N, dtype=torch.int64, device=dev)
    al, cl_ = words // K, L // K
    for idx in (1000, 3):
        q = cl.query(idx)
        for g in range(G):
            srvs[g].set_query(q)
            srvs[g].run_pre()
            for st in range(K):
                srvs[g].first_dim_stage(st)
            srvs[g].sync()
        staged = [a.clone() for a in accs]
        for g in range(G):  # all stages in one launch: the same buffer
            accs[g].zero_()
            srvs[g].first_dim()
            srvs[g].sync()
            assert torch.equal(accs[g], staged[g]), f"first_dim() vs {K} stage launches, rank {g}"
        for st in range(K):  # the per-stage reduce-scatter
            total = torch.stack([a[st * al:(st + 1) * al] for a in accs]).sum(0)
            for g in range(G):
                chunks[g][st * cl_:(st + 1) * cl_].copy_(total[g * cl_:(g + 1) * cl_])
        torch.cuda.synchronize()
        for g in range(G):
            srvs[g].fold_local(chunks[g].data_ptr(), cts[g].data_ptr())
            srvs[g].sync()
        gathered.copy_(torch.cat(cts))
        torch.cuda.synchronize()
        srvs[0].fold_root(gathered.data_ptr())
        srvs[0].sync()
        assert_eq(srvs[0].read(SV.BUF_FINAL), O.answer(po, q, wl, wr, w, v, db), f"pipelined sweep G={G} K={K} idx={idx}")
    with pytest.raises(Exception):
        srvs[0].set_sweep_stages(s.num_per // 16)  # fewer than one 64-column block per stage
    for srv in srvs:
        srv.close()


@pytest.mark.slow
def test_full_size_stream_direct_upload(sa, oracle):
    """SpiralStream-style direct upload at the published "(20, 256)/spiralstream" parameters
    (all_parameter_choices.txt:82-97: nu1=9, nu2=6, p=256, q'=2^19, t_GSW=5, t_conv=4): 512 + 30 uploaded
    Regev ciphertexts, no expansion, 2 GiB NTT-form database on the device.  Property: decodes to the item."""
    O = oracle
    kw = dict(t_gsw=5, t_conv=4, t_exp=2, t_exp_right=56, qprime_bits=19, p_db=256, direct_upload=1)
    po, pg = O.make_params(9, 6, **kw), sa.make_params(9, 6, **kw)
    cl = O.Client(po, seed=4)
    wl, wr, w, v = cl.pub_params()
    srv = sa.Server(pg)
    srv.gen_db(777)
    srv.set_pub_params(wl, wr, w, v)
    for idx in (31337 % (1 << 15), 0):
        fin, resp, us = srv.answer(cl.query(idx))
        assert_eq(cl.decode(resp), O.db_item(po, 777, idx), "decoded plaintext (direct upload, full size)")
    srv.close()


@pytest.mark.parametrize("nu1,nu2,p_db,bits", [(3, 5, 256, 8), (4, 2, 32768, 15), (3, 3, 512, 9), (2, 1, 256, 64), (4, 4, 1 << 20, 20)])
def test_raw_ingest_matches_load_db(sa, oracle, nu1, nu2, p_db, bits, opts):
    """plaintext coefficients in (bit-packed, the item size of select_params.py:297), device database out --
    the centred lift, the transforms and the layout of load_db (src/spiral.cpp:1083-1171) on the device.  Must give the
    database the reference's load_db builds: compared through read_db_slots / read_db_item and through the accumulators.
    Several staging passes, a sharded server fed the whole stream, and an out-of-range coefficient are covered."""
    O = oracle
    from spiral_amd import server as SV

    kw = dict(t_gsw=8, p_db=p_db, qprime_bits=27 if p_db > 4096 else 20)
    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    s = O.shape_of(po)
    total = s.dim0 * s.num_per
    pts = np.stack([O.db_item(po, 41, i) for i in range(total)])
    pts[0, 0, 0, :4] = [0, p_db - 1, p_db // 2, p_db // 2 - 1]  # both sides of the centring threshold
    db = np.zeros(O.db_words(po), dtype=np.uint64).reshape(N, s.num_per, 2, s.dim0, 2)
    for i in range(total):
        enc = O.encode_item(po, pts[i])  # [m][c][limb][z]
        db[:, i % s.num_per, :, i // s.num_per, :] = (enc[:, :, 0, :] | (enc[:, :, 1, :] << np.uint64(32))).transpose(2, 1, 0)
    items = O.pack_items(pts, bits)
    assert items.nbytes == total * 4 * N * bits // 8
    opts(db_stage_bytes=5 * 4 * N * bits // 8)  # 5 items per staging pass
    srv = sa.Server(pg)
    srv.load_db_items(items, bits)
    assert_eq(srv.read_db_slots(0, N), db, "ingested database == load_db's")
    for i in (0, total - 1, total // 2 + 1):
        assert_eq(srv.read_db_item(i), O.encode_item(po, pts[i]), f"item {i}")
    cl = O.Client(po, seed=3)
    wl, wr, w, v = cl.pub_params()
    srv.set_pub_params(wl, wr, w, v)
    q = cl.query(total - 2)
    fin, resp, _ = srv.answer(q)
    assert_eq(fin, O.answer(po, q, wl, wr, w, v, db.reshape(-1)), "answer from the ingested database")
    if p_db <= 512:  # (the larger plaintext moduli are bit-exact too but these toy parameter sets are too noisy to decode them)
        assert_eq(cl.decode(resp), pts[total - 2], "decoded plaintext")
    # a second server on the upper half of the first dimension, fed the whole stream in two calls: keeps its own items
    half = sa.Server(pg, 0, s.dim0 // 2, s.dim0)
    cut = total // 3
    item_bytes = 4 * N * bits // 8
    half.load_db_items(items[:cut * item_bytes], bits, 0, cut)
    half.load_db_items(items[cut * item_bytes:], bits, cut, total - cut)
    assert_eq(half.read_db_slots(3, 2), db[3:5, :, :, s.dim0 // 2:, :], "sharded ingest")
    with pytest.raises(sa.SpiralGpuError):
        half.read_db_item(0)  # not in this shard
    if bits != 64 and (1 << bits) > p_db or bits == 64:
        bad = pts.copy()
        bad[1, 1, 1, 7] = p_db  # the reference asserts val < p_db (src/spiral.cpp:1117)
        with pytest.raises(sa.SpiralGpuError):
            srv.load_db_items(O.pack_items(bad, bits), bits)
    srv.close()
    half.close()


@pytest.mark.parametrize("nu1,nu2,kw,G,graphs", [(4, 4, dict(t_gsw=4), 2, False), (4, 4, dict(t_gsw=4), 4, True), (4, 4, dict(t_gsw=4), 8, False),
                                                  (6, 2, {}, 4, False), (5, 3, dict(t_gsw=8), 8, True), (8, 2, dict(t_gsw=4), 16, False)])
def test_sharded_expansion_emulated_on_one_gpu(sa, oracle, nu1, nu2, kw, G, graphs):
    """every rank expands only the subtree above its own first-dimension block and every G-th GSW bit, the blocks of GSW bits
    are all-gathered (torch.cat stands in for the collective), and from there the answer runs as in the unsharded N-GPU flow:
    ScalToMat outputs of each rank's block, the GSW matrices on every rank and the final answer must equal the oracle's"""
    import torch
    from spiral_amd import server as SV

    O = oracle
    po, pg = O.make_params(nu1, nu2, **kw), sa.make_params(nu1, nu2, **kw)
    s = O.shape_of(po)
    assert s.stopround > 0
    cl = O.Client(po, seed=41)
    wl, wr, w, v = cl.pub_params()
    db = O.gen_db(po, 19)
    dev = torch.device("cuda", 0)
    words = s.num_per * 6 * N
    per = s.dim0 // G
    srvs, accs, blocks = [], [], []
    for g in range(G):
        srv = sa.Server(pg, 0, g * per, (g + 1) * per)
        srv.gen_db(19)
        srv.set_pub_params(wl, wr, w, v)
        srv.set_expand_shard(g, G)
        srv.keep_cts(True)
        accs.append(torch.zeros(words, dtype=torch.int64, device=dev))
        srv.set_acc(accs[g].data_ptr())
        blocks.append(torch.zeros(srv.gsw_bits_words(), dtype=torch.int64, device=dev))
        srv.use_graphs(graphs)
        srvs.append(srv)
    gathered = torch.zeros(G * blocks[0].numel(), dtype=torch.int64, device=dev)
    total = s.dim0 * s.num_per
    for idx in ((total - 1, 5, total // 2) if graphs else (total // 3,)):
        q = cl.query(idx)
        cts, gsw = O.stage_convert(po, O.stage_expand(po, q, wl, wr), w, v)
        for g in range(G):
            srvs[g].set_query(q)
            srvs[g].run_expand_pack(blocks[g].data_ptr())
            srvs[g].sync()
        gathered.copy_(torch.cat(blocks))  # the all-gather
        torch.cuda.synchronize()
        for g in range(G):
            srvs[g].run_unpack_convert_sweep(gathered.data_ptr())
            srvs[g].sync()
            assert_eq(srvs[g].read(SV.BUF_CTS), cts[g * per:(g + 1) * per], f"rank {g}: scalToMat outputs of its own block")
            assert_eq(srvs[g].read(SV.BUF_GSW), gsw, f"rank {g}: regevToGSW outputs")
        srvs[0].set_acc(0)  # root folds alone from the summed accumulators
        tot = torch.stack(accs).sum(0)
        torch.cuda.synchronize()
        srvs[0].set_acc(tot.data_ptr())
        srvs[0].run_post(reduce_first=True)
        srvs[0].sync()
        assert_eq(srvs[0].read(SV.BUF_FINAL), O.answer(po, q, wl, wr, w, v, db), f"answer with sharded expansion, G={G} idx={idx}")
        assert_eq(cl.decode(srvs[0].read(SV.BUF_RESPONSE)), O.db_item(po, 19, idx), "decoded plaintext")
        srvs[0].set_acc(accs[0].data_ptr())
    for srv in srvs:
        srv.close()

"""CPU tests of the parity oracle itself: the reference's own self-checks restated
(do_MatPol_test src/spiral.cpp:1181, "Is correct?" src/spiral.cpp:1412-1494) plus algebraic
properties that pin each restated function independently of the others."""
import numpy as np
import pytest

N = 2048


def br(i):
    return int(format(i, "011b")[::-1], 2)


def test_ntt_is_the_negacyclic_evaluation_map(oracle):
    """forward output slot k (bit-reversed order) = a(psi^(2*bitrev(k)+1)) for the minimal 4096-th roots"""
    O = oracle
    rng = np.random.default_rng(1)
    a = rng.integers(0, O.Q, size=N, dtype=np.uint64)
    f = O.to_ntt(a)
    for n, (m, psi) in enumerate(((O.P, 66687), (O.B, 158221))):
        am = [int(x) % m for x in a]
        for k in [0, 1, 2, 1023, 1024, 2047]:
            e = pow(psi, 2 * br(k) + 1, m)
            s, x = 0, 1
            for j in range(N):
                s = (s + am[j] * x) % m
                x = x * e % m
            assert s == int(f[n, k])


def test_matpoly_roundtrip(oracle):
    """do_MatPol_test: from_ntt(to_ntt(A)) == A for a random 3x6 matrix mod Q"""
    O = oracle
    rng = np.random.default_rng(2)
    A = rng.integers(0, O.Q, size=(3, 6, N), dtype=np.uint64)
    assert (O.from_ntt(O.to_ntt(A)) == A).all()
    # edge values: 0, 1, Q-1, Q (a lazy "Q" must come back as 0)
    e = np.zeros(N, dtype=np.uint64)
    e[0], e[1], e[2], e[3] = 0, 1, O.Q - 1, O.Q
    back = O.from_ntt(O.to_ntt(e))
    assert list(back[:4]) == [0, 1, O.Q - 1, 0]


def test_multiply_is_negacyclic_convolution(oracle):
    O = oracle
    rng = np.random.default_rng(3)
    a = np.zeros(N, dtype=np.uint64)
    b = np.zeros(N, dtype=np.uint64)
    ia = rng.choice(N, 5, replace=False)
    ib = rng.choice(N, 5, replace=False)
    a[ia] = rng.integers(0, O.Q, 5, dtype=np.uint64)
    b[ib] = rng.integers(0, O.Q, 5, dtype=np.uint64)
    got = O.from_ntt(O.multiply(O.to_ntt(a).reshape(1, 1, 2, N), O.to_ntt(b).reshape(1, 1, 2, N)))[0, 0]
    exp = [0] * N
    for i in ia:
        for j in ib:
            pr = int(a[i]) * int(b[j]) % O.Q
            k = int(i + j)
            if k < N:
                exp[k] = (exp[k] + pr) % O.Q
            else:
                exp[k - N] = (exp[k - N] - pr) % O.Q
    assert [int(x) for x in got] == exp


def test_automorph_and_invert_edge_semantics(oracle):
    O = oracle
    a = np.arange(N, dtype=np.uint64) + 5
    a[7] = 0
    t = N // 4 + 1
    out = O.automorph(a, t)
    for i in [0, 1, 7, 100, 2047]:
        pos, wraps = (i * t) % N, (i * t) // N
        assert int(out[pos]) == (O.Q - int(a[i]) if wraps & 1 else int(a[i]))
    assert int(O.invert(np.zeros(N, dtype=np.uint64))[0]) == O.Q  # 0 -> Q, poly.cpp:279


def test_gadget_invert_recomposes(oracle):
    O = oracle
    rng = np.random.default_rng(4)
    for t in (2, 4, 5, 8, 10, 16, 56):
        bits = O.lib().orc_get_bits_per(t)
        assert bits == (1 if t == 56 else 56 // t + 1)
        v = rng.integers(0, O.Q, size=(1, 1, N), dtype=np.uint64)
        v[0, 0, 0] = O.Q  # the Q - 0 case
        d = O.gadget_invert(v, t, 1)
        assert d.max() < (1 << bits)
        rec = sum(d[k, 0].astype(object) << (bits * k) for k in range(t))
        assert (rec == v[0, 0].astype(object)).all()


def test_split_and_crt_is_a_signed_decomposition(oracle):
    O = oracle
    rng = np.random.default_rng(5)
    for t in (4, 5, 8, 10):
        bits = O.lib().orc_get_bits_per(t)
        raw = rng.integers(0, O.Q, size=(2, 3, 2, N), dtype=np.uint64)
        raw[0, 0, 0, :4] = [0, 1, O.Q - 1, (1 << (bits - 1))]
        dig = O.from_ntt(O.split_and_crt(raw, t))  # [2][3t][2][N] digits mod Q
        for r in range(3):
            rec = np.zeros((2, 2, N), dtype=object)
            for k in range(t):
                rec = (rec + (dig[:, r + 3 * k].astype(object) << (bits * k))) % O.Q
            assert (rec == raw[:, r].astype(object) % O.Q).all()
        cent = np.where(dig.astype(object) > O.Q // 2, dig.astype(object) - O.Q, dig.astype(object))
        assert abs(cent).max() <= (1 << bits)


def test_rescale_edges(oracle):
    O = oracle
    qp, q1 = 786433, 1024
    for a in [0, 1, O.Q // 2 - 1, O.Q // 2, O.Q // 2 + 1, O.Q - 1, 12345678901234567]:
        for out_mod in (qp, q1):
            c = a - O.Q if a >= O.Q // 2 else a
            num = c * out_mod
            sign = 1 if c >= 0 else -1
            val = num + sign * (O.Q // 2)
            res = abs(val) // O.Q * (1 if val >= 0 else -1)  # C truncating division
            assert O.rescale(a, O.Q, out_mod) == res % out_mod


def test_sweep_matches_matrix_product(oracle):
    """first-dimension sweep == sum_j ct_j * pt_{i,j} computed with multiply() on the unpacked operands"""
    O = oracle
    p = O.make_params(2, 1)
    s = O.shape_of(p)
    rng = np.random.default_rng(6)
    cts = np.stack([rng.integers(0, m, size=(s.dim0, 3, 2, N), dtype=np.uint64) for m in (O.P, O.B)], axis=3)
    db = O.gen_db(p, 99)
    got = O.multiply_query_by_database(O.reorient_ciphertexts(cts), db, s.dim0, s.num_per)
    dbv = db.reshape(N, s.num_per, 2, s.dim0, 2)  # z, ii, c, j, m
    for ii in range(s.num_per):
        acc = np.zeros((3, 2, 2, N), dtype=object)
        for j in range(s.dim0):
            pt = np.zeros((2, 2, 2, N), dtype=np.uint64)  # m, c, limb, z
            w = dbv[:, ii, :, j, :]  # z, c, m
            pt[:, :, 0, :] = (w & 0xFFFFFFFF).transpose(2, 1, 0)
            pt[:, :, 1, :] = (w >> 32).transpose(2, 1, 0)
            acc = acc + O.multiply(cts[j], pt).astype(object)
        acc[:, :, 0] %= O.P
        acc[:, :, 1] %= O.B
        assert (acc == got[ii].astype(object)).all()


@pytest.mark.parametrize(
    "nu1,nu2,kw",
    [
        (2, 1, {}),  # stopround == 0 branch
        (4, 2, dict(t_gsw=4)),  # stopround > 0 branch
        (3, 3, dict(t_gsw=8)),
        (2, 2, dict(t_gsw=5, t_exp=2, qprime_bits=19, direct_upload=1)),  # SpiralStream-style direct upload
    ],
)
def test_end_to_end_is_correct(oracle, nu1, nu2, kw):
    """the reference's only functional oracle: decode(answer(query(idx))) == DB[idx]"""
    O = oracle
    p = O.make_params(nu1, nu2, **kw)
    s = O.shape_of(p)
    db = O.gen_db(p, 1234)
    c = O.Client(p, seed=11)
    wl, wr, w, v = c.pub_params()
    total = s.dim0 * s.num_per
    for idx in {0, total - 1, 1234 % total}:
        fin = O.answer(p, c.query(idx), wl, wr, w, v, db)
        pt = c.decode(O.stage_rescale(p, fin))
        assert (pt == O.db_item(p, 1234, idx)).all(), idx


def test_staged_pipeline_equals_answer(oracle):
    O = oracle
    p = O.make_params(3, 2, t_gsw=4)
    db = O.gen_db(p, 5)
    c = O.Client(p, seed=3)
    wl, wr, w, v = c.pub_params()
    q = c.query(9)
    cv = O.stage_expand(p, q, wl, wr)
    cts, gsw = O.stage_convert(p, cv, w, v)
    raw = O.stage_first_dim(p, cts, db)
    fin = O.stage_fold(p, raw, gsw)
    assert (fin == O.answer(p, q, wl, wr, w, v, db)).all()


@pytest.mark.parametrize("dim0", [1, 2, 16, 32, 64, 128])
def test_vectorised_sweep_cell_equals_scalar(oracle, oracle_mt, dim0):
    """the first-dimension cell in the reference's AVX2 (default build) and AVX-512 (native build, where the host has it) forms
    (src/spiral.cpp:640-886: partial reduction every 64 terms) against the scalar u128 cell: random residues, all-maximal
    residues (the largest sums) and zeros; dim0 = 1 takes the scalar cell in every build (odd pairing)"""
    rng = np.random.default_rng(dim0)
    num_per = 3
    for O in (oracle, oracle_mt):
        for fill in ("random", "max", "zero"):
            def words(shape):
                if fill == "random":
                    lo, hi = rng.integers(0, O.P, size=shape, dtype=np.uint64), rng.integers(0, O.B, size=shape, dtype=np.uint64)
                elif fill == "max":
                    lo, hi = np.full(shape, O.P - 1, dtype=np.uint64), np.full(shape, O.B - 1, dtype=np.uint64)
                else:
                    lo = hi = np.zeros(shape, dtype=np.uint64)
                return np.ascontiguousarray(lo | (hi << np.uint64(32)))
            re = words((O.N, dim0, 2, 4))
            db = words((O.N, num_per, 2, dim0, 2))
            got = O.multiply_query_by_database(re, db, dim0, num_per)
            want = O.multiply_query_by_database_scalar(re, db, dim0, num_per)
            assert (got == want).all(), (O.sweep_isa(), fill, dim0)
            if fill == "max" and dim0 >= 32:
                assert int(want.max()) > 0
    assert oracle.sweep_isa() in ("avx2", "avx512", "scalar") and oracle_mt.sweep_isa() in ("avx2", "avx512", "scalar")


def test_vectorised_transforms_equal_scalar(oracle, oracle_mt):
    """the transforms in the reference's USE_AVX2 form (forward butterflies four at a time for t >= 4, scalar below; both closing
    corrections as vector compares, src/core.cpp:292-349, 479-506) against the scalar restatement: random residues, lazy inputs up to
    4m - 1 (ntt_forward accepts them), the extreme patterns 0 / m - 1 / 2m / 2m - 1 / 4m - 1 in every slot, for both builds; and a
    whole answer computed with the vector transforms switched on is word for word the scalar one"""
    rng = np.random.default_rng(77)
    for O in (oracle, oracle_mt):
        if O.ntt_isa() != "avx2":
            pytest.skip("oracle built without AVX2")
        mods = (O.P, O.B)
        cases = [np.stack([rng.integers(0, m, size=(4, O.N), dtype=np.uint64) for m in mods], axis=1),
                 np.stack([rng.integers(0, 4 * m, size=(4, O.N), dtype=np.uint64) for m in mods], axis=1)]
        for f in (lambda m: 0, lambda m: m - 1, lambda m: 2 * m, lambda m: 2 * m - 1, lambda m: 4 * m - 1, lambda m: m):
            cases.append(np.stack([np.full((1, O.N), f(m), dtype=np.uint64) for m in mods], axis=1))
        try:
            for x in cases:
                want_f = O.ntt_forward_scalar(x)
                canon = np.stack([x[:, i] % np.uint64(m) for i, m in enumerate(mods)], axis=1)  # the inverse takes values below 2m
                want_i = O.ntt_inverse_scalar(canon)
                assert O.set_ntt_simd(True) is True
                assert (O.ntt_forward(x) == want_f).all() and (O.ntt_inverse(canon) == want_i).all()
                assert O.set_ntt_simd(False) is False
                assert (O.ntt_forward(x) == want_f).all()
            po = O.make_params(3, 3, t_gsw=4)
            cl = O.Client(po, seed=8)
            wl, wr, w, v = cl.pub_params()
            db, q = O.gen_db(po, 2), cl.query(37)
            want = O.answer(po, q, wl, wr, w, v, db)
            O.set_ntt_simd(True)
            assert (O.answer(po, q, wl, wr, w, v, db) == want).all()
        finally:
            O.set_ntt_simd(False)


def test_threaded_native_build_gives_identical_results(oracle, oracle_mt):
    """the -fopenmp -march=native build on many threads (full-size parity tests, bench.py's all-cores baseline) must be
    the same function as the default single-threaded build: base path with expansion + stopround, and the pack path"""
    O, M = oracle, oracle_mt
    assert M.n_threads >= 1
    kw = dict(t_gsw=4)
    po = O.make_params(4, 3, **kw)
    cl = O.Client(po, seed=12)
    wl, wr, w, v = cl.pub_params()
    q = cl.query(77)
    db = O.gen_db(po, 5)
    assert np.array_equal(M.gen_db(M.make_params(4, 3, **kw), 5), db)
    assert np.array_equal(M.answer(M.make_params(4, 3, **kw), q, wl, wr, w, v, db), O.answer(po, q, wl, wr, w, v, db))
    pk = dict(t_gsw=4)
    pp = O.make_params(5, 2, **pk)
    pc = O.PackClient(pp, 2, seed=3)
    pwl, pwr, pv, pvw = pc.pub_params()
    pq = pc.query(9)
    pdb = O.pack_gen_db(pp, 2, 31)
    assert np.array_equal(M.pack_gen_db(M.make_params(5, 2, **pk), 2, 31), pdb)
    r0, f0 = O.pack_answer(pp, 2, pq, pwl, pwr, pv, pvw, pdb)
    r1, f1 = M.pack_answer(M.make_params(5, 2, **pk), 2, pq, pwl, pwr, pv, pvw, pdb)
    assert np.array_equal(r0, r1) and np.array_equal(f0, f1)
    # the slot-subset sweep is the full sweep restricted to those slots
    rng = np.random.default_rng(3)
    dim0, num_per = 8, 4
    cts = np.stack([rng.integers(0, m, size=(dim0, 3, 2, O.N), dtype=np.uint64) for m in (O.P, O.B)], axis=-2)
    re = O.reorient_ciphertexts(cts)
    dbr = O.fill_db_random(1, dim0 * num_per * 4 * O.N)
    full = O.multiply_query_by_database(re, dbr, dim0, num_per)
    zs = [0, 5, 2047, 1024]
    sub = O.multiply_query_by_database_slots(re[zs], dbr.reshape(O.N, -1)[zs], dim0, num_per)
    assert np.array_equal(sub, full[..., zs])


def test_response_wire_form_round_trip_and_host_unpack(oracle):
    """the bit-packed response (write_arbitrary_bits along modswitch's walk, src/core.cpp:20-52, src/spiral.cpp:40-76) at the two widths
    of the summary's response size: size == the reference's formula (20 480 B at config 2), fields at their bit offsets, the
    oracle's and the library's host-side unpack both invert it"""
    import spiral_amd as sa

    O = oracle
    rng = np.random.default_rng(3)
    assert O.response_wire_bytes(O.make_params(8, 7)) == 20480  # response 20 480 B at (20, 256)
    for kw in (dict(), dict(qprime_bits=27, p_db=32768), dict(qprime_bits=36, p_db=8388592), dict(qprime_bits=14, p_db=2), dict(qprime_bits=31, p_db=524288)):
        for out_n in (2, 3, 12):
            po, pg = O.make_params(8, 7, **kw), sa.make_params(8, 7, **kw)
            w0, w1 = po.qprime_bits, int(np.ceil(np.log2(4 * po.p_db)))
            resp = np.zeros((out_n + 1, out_n, N), dtype=np.uint64)
            resp[0] = rng.integers(0, 1 << w0, size=(out_n, N), dtype=np.uint64)
            resp[1:] = rng.integers(0, 4 * po.p_db, size=(out_n, out_n, N), dtype=np.uint64)
            resp[0, 0, :2] = [(1 << w0) - 1, 0]
            resp[-1, -1, -1] = 4 * po.p_db - 1
            wire = O.response_to_wire(po, resp, out_n)
            assert len(wire) == (out_n * N * w0 + out_n * out_n * N * w1) // 8 == sa.response_wire_bytes(pg, out_n)
            bits = np.unpackbits(wire, bitorder="little")
            val = lambda off, w: int(sum(int(b) << i for i, b in enumerate(bits[off:off + w])))
            assert val(0, w0) == (1 << w0) - 1 and val(w0, w0) == 0 and val(7 * w0, w0) == int(resp[0, 0, 7])
            assert val(out_n * N * w0 + 5 * w1, w1) == int(resp[1, 0, 5]) and val(len(bits) - w1, w1) == 4 * po.p_db - 1
            assert (O.response_from_wire(po, wire, out_n) == resp).all()
            assert (sa.response_from_wire(pg, wire, out_n) == resp).all()

"""The oracle's SpiralPack restatement against the reference's own functional check for that path:
"Is correct? :" of src/testing.cpp:1136 -- query, answer (expansion or direct upload, first dimension,
folding, packing, modulus switch) and decode must return the out_n x out_n database items."""
import numpy as np
import pytest


@pytest.mark.parametrize(
    "nu1,nu2,out_n,kw",
    [
        (6, 2, 2, {}),  # the probe run of SURVEY.md 8c: ./spiral 6 2 77 a --high-rate
        (5, 2, 3, dict(t_gsw=4)),
        (3, 2, 2, dict(t_gsw=5, t_exp=2, qprime_bits=19, direct_upload=1)),  # SpiralStreamPack: uploaded cts
    ],
)
def test_pack_end_to_end_is_correct(oracle, nu1, nu2, out_n, kw):
    O = oracle
    p = O.make_params(nu1, nu2, **kw)
    s = O.pack_shape_of(p, out_n)
    db = O.pack_gen_db(p, out_n, 31)
    c = O.PackClient(p, out_n, seed=8)
    wl, wr, v, vw = c.pub_params()
    total = s.dim0 * s.num_per
    for idx in {77 % total, total - 1}:
        resp, _ = O.pack_answer(p, out_n, c.query(idx), wl, wr, v, vw, db)
        assert (c.decode(resp) == O.pack_db_item(p, out_n, 31, idx)).all(), idx


def test_pack_sweep_is_the_matrix_product(oracle):
    O = oracle
    rng = np.random.default_rng(3)
    dim0, num_per = 4, 2
    cts = np.stack([rng.integers(0, m, size=(dim0, 2, O.N), dtype=np.uint64) for m in (O.P, O.B)], axis=2)
    db = O.fill_db_random(5, dim0 * num_per * O.N)
    got = O.sweep_dim1(db, O.reorient_dim1(cts, dim0, 1), dim0, num_per)
    dbv = db.reshape(O.N, num_per, dim0)
    for ii in range(num_per):
        acc = np.zeros((2, 2, O.N), dtype=object)
        for j in range(dim0):
            w = dbv[:, ii, j]
            acc[:, 0] += cts[j, :, 0].astype(object) * (w & 0xFFFFFFFF).astype(object)
            acc[:, 1] += cts[j, :, 1].astype(object) * (w >> 32).astype(object)
        acc[:, 0] %= O.P
        acc[:, 1] %= O.B
        assert (acc == got[ii].astype(object)).all()

"""The oracle's regenerated twiddle tables and constants vs the reference's own data
(tests/golden/ntt_tables.json, produced by tests/golden/make_golden.py from src/constants.cpp:16 and
include/values.h)."""
import hashlib
import json
import os

import numpy as np

GOLD = json.load(open(os.path.join(os.path.dirname(__file__), "golden", "ntt_tables.json")))
NAMES = ["inv_p_w", "inv_p_wscaled", "inv_b_w", "inv_b_wscaled", "fwd_p_w", "fwd_p_wscaled", "fwd_b_w", "fwd_b_wscaled"]


def test_tables_match_reference_data(oracle):
    t = oracle.get_tables()
    for r, name in enumerate(NAMES):
        row = np.ascontiguousarray(t[r], dtype="<u8")
        assert hashlib.sha256(row.tobytes()).hexdigest() == GOLD["rows"][name]["sha256"], name
        for i, v in GOLD["rows"][name]["samples"].items():
            assert int(row[int(i)]) == v


def test_constants_match_reference(oracle):
    c = GOLD["constants"]
    assert (c["p_i"], c["b_i"]) == (oracle.P, oracle.B)
    assert (c["n0"], c["n1"], c["n2"]) == (2, 3, 2)
    # Barrett ratios of values.h are floor(2^64/m) and floor(2^128/Q): `%` in the oracle is the same function
    assert c["cr1_p"] == (1 << 64) // oracle.P and c["cr1_b"] == (1 << 64) // oracle.B
    assert (c["cr1_Q"] << 64) + c["cr0_Q"] == (1 << 128) // oracle.Q
    # CRT lift constants (values.h:24-25)
    assert c["b_inv_pa_factor"] == pow(oracle.B, -1, oracle.P) and c["pa_inv_b_factor"] == pow(oracle.P, -1, oracle.B)
    for x, y in [(0, 0), (1, 0), (0, 1), (12345, 54321), (oracle.P - 1, oracle.B - 1), (oracle.P, oracle.B)]:
        v = oracle.lib().orc_crt_compose(x, y)
        assert v < oracle.Q and v % oracle.P == x % oracle.P and v % oracle.B == y % oracle.B
    for bits, q in enumerate(c["qprime_mods"]):
        if q:
            p = oracle.make_params(2, 1, qprime_bits=bits)
            assert oracle.shape_of(p).qprime == q

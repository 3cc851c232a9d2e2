"""Function-level mirror of the reference's hot-path interface (host numpy buffers in reference
layouts), each a direct call through the C ABI.  See include/spiral_gpu.h for the reference citations."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import U64P, Params, Shape, check, lib

N = 2048
P = 268369921
B = 249561089
Q = P * B

__all__ = [
    "N", "P", "B", "Q", "make_params", "get_shape", "get_tables", "ntt_forward", "ntt_inverse", "to_ntt", "to_ntt_no_reduce", "from_ntt",
    "multiply", "add", "mul_by_const", "automorph", "invert", "gadget_invert", "getRescaled", "multiplyQueryByDatabase", "multiplyQueriesByDatabase", "split_and_crt",
    "foldOneFurtherDimension", "expandImproved", "scalToMat", "regevToGSW", "time_ntt", "time_ntt_digits", "response_wire_bytes", "response_from_wire",
    "set_option", "get_option", "options",
]


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(U64P)


def _c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def set_option(name: str, value: int) -> None:
    """spiral_gpu_set_option: a process-wide schedule option (include/spiral_gpu.h lists them); servers take the values in force at creation"""
    check(lib().spiral_gpu_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = C.c_int64()
    check(lib().spiral_gpu_get_option(name.encode(), C.byref(v)))
    return v.value


class options:
    """with sa.options(fold_pair=0, fwd2=1): ...  -- sets the options and restores the previous values on exit"""

    def __init__(self, **kw):
        self.kw = kw

    def __enter__(self):
        self.old = {k: get_option(k) for k in self.kw}
        for k, v in self.kw.items():
            set_option(k, v)
        return self

    def __exit__(self, *exc):
        for k, v in self.old.items():
            set_option(k, v)
        return False


def make_params(nu1, nu2, t_gsw=8, t_conv=4, t_exp=8, t_exp_right=56, qprime_bits=20, p_db=256, direct_upload=0) -> Params:
    return Params(nu1, nu2, t_gsw, t_conv, t_exp, t_exp_right, qprime_bits, direct_upload, p_db)


def get_shape(p: Params) -> Shape:
    s = Shape()
    check(lib().spiral_gpu_get_shape(C.byref(p), C.byref(s)))
    return s


def get_tables() -> np.ndarray:
    out = np.zeros((8, N), dtype=np.uint64)
    check(lib().spiral_gpu_get_tables(_p(out)))
    return out


def ntt_forward(operand) -> np.ndarray:
    """ntt_forward (src/core.cpp:247) on [..., 2, N]; returns a new array"""
    x = _c(operand).copy()
    check(lib().spiral_gpu_ntt_forward(_p(x), x.size // (2 * N)))
    return x


def ntt_inverse(operand) -> np.ndarray:
    x = _c(operand).copy()
    check(lib().spiral_gpu_ntt_inverse(_p(x), x.size // (2 * N)))
    return x


def to_ntt(raw) -> np.ndarray:
    raw = _c(raw)
    out = np.zeros(raw.shape[:-1] + (2, N), dtype=np.uint64)
    check(lib().spiral_gpu_to_ntt(_p(out), _p(raw), raw.size // N, 1))
    return out


def to_ntt_no_reduce(raw) -> np.ndarray:
    raw = _c(raw)
    out = np.zeros(raw.shape[:-1] + (2, N), dtype=np.uint64)
    check(lib().spiral_gpu_to_ntt(_p(out), _p(raw), raw.size // N, 0))
    return out


def from_ntt(a) -> np.ndarray:
    a = _c(a)
    out = np.zeros(a.shape[:-2] + (N,), dtype=np.uint64)
    check(lib().spiral_gpu_from_ntt(_p(out), _p(a), a.size // (2 * N)))
    return out


def multiply(a, b) -> np.ndarray:
    a, b = _c(a), _c(b)
    rs, ms, cs = a.shape[0], a.shape[1], b.shape[1]
    assert b.shape[0] == ms
    out = np.zeros((rs, cs, 2, N), dtype=np.uint64)
    check(lib().spiral_gpu_multiply(_p(out), _p(a), _p(b), rs, ms, cs))
    return out


def add(a, b) -> np.ndarray:
    a, b = _c(a), _c(b)
    out = np.zeros_like(a)
    check(lib().spiral_gpu_add(_p(out), _p(a), _p(b), a.size // (2 * N)))
    return out


def mul_by_const(single, a) -> np.ndarray:
    single, a = _c(single), _c(a)
    out = np.zeros_like(a)
    check(lib().spiral_gpu_mul_by_const(_p(out), _p(single), _p(a), a.size // (2 * N)))
    return out


def automorph(raw, t) -> np.ndarray:
    raw = _c(raw)
    out = np.zeros_like(raw)
    check(lib().spiral_gpu_automorph(_p(out), _p(raw), raw.size // N, int(t)))
    return out


def invert(raw) -> np.ndarray:
    raw = _c(raw)
    out = np.zeros_like(raw)
    check(lib().spiral_gpu_invert(_p(out), _p(raw), raw.size // N))
    return out


def gadget_invert(raw, mx, rdim) -> np.ndarray:
    raw = _c(raw)
    cols = raw.shape[1]
    out = np.zeros((mx, cols, N), dtype=np.uint64)
    check(lib().spiral_gpu_gadget_invert(_p(out), _p(raw), mx, rdim, cols))
    return out


def getRescaled(a, inp_mod, out_mod) -> np.ndarray:
    a = _c(a)
    out = np.zeros_like(a)
    check(lib().spiral_gpu_get_rescaled(_p(out), _p(a), a.size, int(inp_mod), int(out_mod)))
    return out


def response_wire_bytes(params, out_n: int = 2) -> int:
    """size of a response's wire form (include/spiral_gpu.h): the "Response size" of the reference's summary"""
    return int(lib().spiral_gpu_response_wire_bytes(C.byref(params), out_n))


def response_from_wire(params, wire, out_n: int = 2) -> np.ndarray:
    """client half of the wire form (load_modswitched_into_ct, src/client.cpp:90): bytes -> [(out_n+1)][out_n][N] values; host code"""
    wire = np.ascontiguousarray(wire, dtype=np.uint8)
    assert wire.size >= response_wire_bytes(params, out_n)
    out = np.zeros((out_n + 1, out_n, N), dtype=np.uint64)
    check(lib().spiral_gpu_response_from_wire(C.byref(params), out_n, wire.ctypes.data_as(C.c_void_p), _p(out)))
    return out


def multiplyQueryByDatabase(reoriented_cts, database, dim0, num_per) -> np.ndarray:
    out = np.zeros((num_per, 3, 2, 2, N), dtype=np.uint64)
    check(lib().spiral_gpu_multiply_query_by_database(_p(out), _p(_c(reoriented_cts)), _p(_c(database)), dim0, num_per))
    return out


def multiplyQueriesByDatabase(reoriented_cts_list, database, dim0, num_per) -> np.ndarray:
    """n <= 8 queries against one pass over the database (the matrix-core sweep where the geometry allows): [n][num_per][3][2][2][N]"""
    n = len(reoriented_cts_list)
    re = np.ascontiguousarray(np.stack([_c(r).reshape(-1) for r in reoriented_cts_list]))
    out = np.zeros((n, num_per, 3, 2, 2, N), dtype=np.uint64)
    check(lib().spiral_gpu_multiply_queries_by_database(_p(out), _p(re), n, _p(_c(database)), dim0, num_per))
    return out


def split_and_crt(raw_cts, t_gsw) -> np.ndarray:
    raw_cts = _c(raw_cts)
    num_per = raw_cts.shape[0]
    out = np.zeros((num_per, 3 * t_gsw, 2, 2, N), dtype=np.uint64)
    check(lib().spiral_gpu_split_and_crt(_p(out), _p(raw_cts), num_per, t_gsw))
    return out


def foldOneFurtherDimension(cts, num_per, query_ct, query_ct_neg, t_gsw) -> np.ndarray:
    """returns the num_per folded raw ciphertexts (the reference overwrites the first half in place)"""
    x = _c(cts).copy()
    check(lib().spiral_gpu_fold_one_further_dimension(_p(x), num_per, _p(_c(query_ct)), _p(_c(query_ct_neg)), t_gsw))
    return x[:num_per]


def expandImproved(cv_v, g, m_exp, W_left, W_right, m_exp_right, n_right, max_bits_to_gen_right, stopround) -> np.ndarray:
    x = _c(cv_v).copy()
    check(lib().spiral_gpu_expand_improved(_p(x), g, m_exp, _p(_c(W_left)), m_exp_right, _p(_c(W_right)), n_right, max_bits_to_gen_right, stopround))
    return x


def scalToMat(m_conv, cv, W) -> np.ndarray:
    out = np.zeros((3, 2, 2, N), dtype=np.uint64)
    check(lib().spiral_gpu_scal_to_mat(_p(out), _p(_c(cv)), _p(_c(W)), m_conv))
    return out


def regevToGSW(m_conv, t, cv_v, W, V) -> np.ndarray:
    out = np.zeros((3, 3 * t, 2, N), dtype=np.uint64)
    check(lib().spiral_gpu_regev_to_gsw(_p(out), _p(_c(cv_v)), _p(_c(W)), _p(_c(V)), m_conv, t))
    return out


def time_ntt(npolys: int, iters: int = 10):
    """average ms of one batched to_ntt / from_ntt launch over npolys polynomials resident in HBM -> (fwd_ms, inv_ms)"""
    f, i = C.c_float(), C.c_float()
    check(lib().spiral_gpu_time_ntt(npolys, iters, C.byref(f), C.byref(i)))
    return f.value, i.value


def time_ntt_digits(npolys: int, n_digits: int, iters: int = 10) -> float:
    """ms per launch of npolys * n_digits gadget-digit transforms (the launch the conversion / expansion stages are made of)"""
    ms = C.c_float()
    check(lib().spiral_gpu_time_ntt_digits(npolys, n_digits, iters, C.byref(ms)))
    return ms.value

"""ctypes loader for the parity oracle (oracle/liboracle.so).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this module; the product
package (spiral_amd) never does.  Buffers are numpy uint64 arrays in the reference's layouts.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liboracle.so")

N = 2048
P = 268369921
B = 249561089
Q = P * B
NTTP = 2 * N  # words per NTT-form polynomial


class Params(C.Structure):
    _fields_ = [
        ("nu1", C.c_uint32),
        ("nu2", C.c_uint32),
        ("t_gsw", C.c_uint32),
        ("t_conv", C.c_uint32),
        ("t_exp", C.c_uint32),
        ("t_exp_right", C.c_uint32),
        ("qprime_bits", C.c_uint32),
        ("direct_upload", C.c_uint32),
        ("p_db", C.c_uint64),
    ]


class Shape(C.Structure):
    _fields_ = [
        ("dim0", C.c_uint32),
        ("num_per", C.c_uint32),
        ("ell", C.c_uint32),
        ("m2", C.c_uint32),
        ("g", C.c_uint32),
        ("stopround", C.c_uint32),
        ("n_left", C.c_uint32),
        ("n_right", C.c_uint32),
        ("n_query_cts", C.c_uint32),
        ("n_bits", C.c_uint32),
        ("qprime", C.c_uint64),
    ]


def build(force: bool = False) -> str:
    srcs = [os.path.join(HERE, f) for f in ("spiral_oracle.c", "spiral_oracle_pack.c", "spiral_oracle.h")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(s) for s in srcs):
        subprocess.check_call(["make", "-C", HERE, "-s"])
    return LIB_PATH


_lib = None
_u64p = C.POINTER(C.c_uint64)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            build()
        _lib = C.CDLL(LIB_PATH)
        _lib.orc_client_new.restype = C.c_void_p
        _lib.orc_client_new.argtypes = [C.POINTER(Params), C.c_uint64, C.c_int]
        _lib.orc_client_free.argtypes = [C.c_void_p]
        for name in ("orc_words_w_left", "orc_words_w_right", "orc_words_w", "orc_words_v", "orc_words_query"):
            getattr(_lib, name).restype = C.c_size_t
            getattr(_lib, name).argtypes = [C.POINTER(Params)]
        _lib.orc_response_wire_bytes.restype = C.c_size_t
        _lib.orc_response_wire_bytes.argtypes = [C.POINTER(Params), C.c_uint32]
        _lib.orc_rescale.restype = C.c_uint64
        _lib.orc_rescale.argtypes = [C.c_uint64] * 3
        _lib.orc_crt_compose.restype = C.c_uint64
        _lib.orc_crt_compose.argtypes = [C.c_uint64] * 2
        _lib.orc_get_bits_per.restype = C.c_uint32
        _lib.orc_get_bits_per.argtypes = [C.c_uint32]
        _lib.orc_db_coeff.restype = C.c_uint64
        _lib.orc_db_coeff.argtypes = [C.c_uint64] * 4
        _lib.orc_pack_client_new.restype = C.c_void_p
        _lib.orc_pack_client_new.argtypes = [C.POINTER(Params), C.c_uint32, C.c_uint64, C.c_int]
        _lib.orc_pack_client_free.argtypes = [C.c_void_p]
        _lib.orc_pack_words_v.restype = C.c_size_t
        _lib.orc_pack_words_v.argtypes = [C.POINTER(Params)]
        for name in ("orc_pack_words_vw", "orc_pack_words_query"):
            getattr(_lib, name).restype = C.c_size_t
            getattr(_lib, name).argtypes = [C.POINTER(Params), C.c_uint32]
    return _lib


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_u64p)


def u64(*shape) -> np.ndarray:
    return np.zeros(shape, dtype=np.uint64)


def shape_of(p: Params) -> Shape:
    s = Shape()
    if lib().orc_get_shape(C.byref(p), C.byref(s)) != 0:
        raise ValueError("unsupported parameter set")
    return s


def make_params(nu1, nu2, t_gsw=8, t_conv=4, t_exp=8, t_exp_right=56, qprime_bits=20, p_db=256, direct_upload=0) -> Params:
    return Params(nu1, nu2, t_gsw, t_conv, t_exp, t_exp_right, qprime_bits, direct_upload, p_db)


# ---- L1/L2 ----
def get_tables() -> np.ndarray:
    out = u64(8, N)
    lib().orc_get_tables(_p(out))
    return out


def ntt_forward(x: np.ndarray) -> np.ndarray:
    y = np.ascontiguousarray(x, dtype=np.uint64).copy()
    flat = y.reshape(-1, NTTP)
    for k in range(flat.shape[0]):
        lib().orc_ntt_forward(flat[k].ctypes.data_as(_u64p))
    return y


def ntt_inverse(x: np.ndarray) -> np.ndarray:
    y = np.ascontiguousarray(x, dtype=np.uint64).copy()
    flat = y.reshape(-1, NTTP)
    for k in range(flat.shape[0]):
        lib().orc_ntt_inverse(flat[k].ctypes.data_as(_u64p))
    return y


def to_ntt(raw: np.ndarray, reduce: bool = True) -> np.ndarray:
    raw = np.ascontiguousarray(raw, dtype=np.uint64)
    npolys = raw.size // N
    out = u64(*raw.shape[:-1], 2, N)
    (lib().orc_to_ntt if reduce else lib().orc_to_ntt_no_reduce)(_p(out), _p(raw), C.c_size_t(npolys))
    return out


def from_ntt(a: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.uint64)
    npolys = a.size // NTTP
    out = u64(*a.shape[:-2], N)
    lib().orc_from_ntt(_p(out), _p(a), C.c_size_t(npolys))
    return out


def multiply(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    rs, ms = a.shape[0], a.shape[1]
    cs = b.shape[1]
    assert b.shape[0] == ms
    out = u64(rs, cs, 2, N)
    lib().orc_multiply(_p(out), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), C.c_size_t(rs), C.c_size_t(ms), C.c_size_t(cs))
    return out


def add(a, b):
    out = np.zeros_like(a)
    lib().orc_add(_p(out), _p(np.ascontiguousarray(a)), _p(np.ascontiguousarray(b)), C.c_size_t(a.size // NTTP))
    return out


def mul_by_const(single, a):
    out = np.zeros_like(a)
    lib().orc_mul_by_const(_p(out), _p(np.ascontiguousarray(single)), _p(np.ascontiguousarray(a)), C.c_size_t(a.size // NTTP))
    return out


def automorph(raw, t):
    raw = np.ascontiguousarray(raw, dtype=np.uint64)
    out = np.zeros_like(raw)
    lib().orc_automorph(_p(out), _p(raw), C.c_size_t(raw.size // N), C.c_uint64(t))
    return out


def invert(raw):
    raw = np.ascontiguousarray(raw, dtype=np.uint64)
    out = np.zeros_like(raw)
    lib().orc_invert(_p(out), _p(raw), C.c_size_t(raw.size // N))
    return out


def gadget_invert(raw, mx, rdim):
    """raw: [rdim][cols][N] -> [mx][cols][N]"""
    raw = np.ascontiguousarray(raw, dtype=np.uint64)
    cols = raw.shape[1]
    out = u64(mx, cols, N)
    lib().orc_gadget_invert(_p(out), _p(raw), C.c_size_t(mx), C.c_size_t(rdim), C.c_size_t(cols))
    return out


def build_gadget(rows, cols):
    out = u64(rows, cols, N)
    lib().orc_build_gadget(_p(out), C.c_size_t(rows), C.c_size_t(cols))
    return out


def rescale(a, inp_mod, out_mod):
    return lib().orc_rescale(int(a), int(inp_mod), int(out_mod))


def response_wire_bytes(p, out_n=2):
    return int(lib().orc_response_wire_bytes(C.byref(p), out_n))


def response_to_wire(p, resp, out_n=2) -> np.ndarray:
    """switched response [(out_n+1)][out_n][N] -> its wire bytes (modswitch's walk + write_arbitrary_bits at the two widths)"""
    resp = np.ascontiguousarray(resp, dtype=np.uint64)
    n = response_wire_bytes(p, out_n)
    buf = u64(n // 8 + 1)
    lib().orc_response_to_wire(C.byref(p), C.c_uint32(out_n), _p(resp), _p(buf))
    return buf.view(np.uint8)[:n].copy()


def response_from_wire(p, wire, out_n=2) -> np.ndarray:
    n = response_wire_bytes(p, out_n)
    buf = u64(n // 8 + 1)
    buf.view(np.uint8)[:n] = np.ascontiguousarray(wire, dtype=np.uint8)[:n]
    out = u64(out_n + 1, out_n, N)
    lib().orc_response_from_wire(C.byref(p), C.c_uint32(out_n), _p(buf), _p(out))
    return out


# ---- L5 ----
def split_and_crt(raw_cts, t_gsw):
    raw_cts = np.ascontiguousarray(raw_cts, dtype=np.uint64)
    num_per = raw_cts.shape[0]
    out = u64(num_per, 3 * t_gsw, 2, 2, N)
    lib().orc_split_and_crt(_p(out), _p(raw_cts), C.c_size_t(num_per), C.c_uint32(t_gsw))
    return out


def reorient_ciphertexts(cts):
    cts = np.ascontiguousarray(cts, dtype=np.uint64)
    dim0 = cts.shape[0]
    out = u64(N, dim0, 2, 4)
    lib().orc_reorient_ciphertexts(_p(out), _p(cts), C.c_size_t(dim0))
    return out


def multiply_query_by_database(reoriented, db, dim0, num_per):
    out = u64(num_per, 3, 2, 2, N)
    lib().orc_multiply_query_by_database(_p(out), _p(reoriented), _p(db), C.c_size_t(dim0), C.c_size_t(num_per))
    return out


def multiply_query_by_database_scalar(reoriented, db, dim0, num_per):
    """the scalar cell only (the default entry point uses the reference's AVX-512 / AVX2 form where the build has it)"""
    out = u64(num_per, 3, 2, 2, N)
    lib().orc_multiply_query_by_database_scalar(_p(out), _p(reoriented), _p(db), C.c_size_t(dim0), C.c_size_t(num_per))
    return out


def sweep_isa():
    f = lib().orc_sweep_isa
    f.restype = C.c_char_p
    return f().decode()


def multiply_query_by_database_slots(reoriented_slabs, db_slabs, dim0, num_per):
    """the sweep on a subset of NTT slots: reoriented_slabs [nz][dim0][2][4], db_slabs [nz][num_per*2*dim0*2] -> [num_per][3][2][2][nz]"""
    nz = reoriented_slabs.shape[0]
    out = u64(num_per, 3, 2, 2, nz)
    lib().orc_multiply_query_by_database_slots(_p(out), _p(np.ascontiguousarray(reoriented_slabs)), _p(np.ascontiguousarray(db_slabs)), C.c_size_t(dim0),
                                               C.c_size_t(num_per), C.c_uint32(nz))
    return out


def expand_improved(cv, g, t_exp, w_left, t_exp_right, w_right, n_right, max_bits_right, stopround):
    cv = np.ascontiguousarray(cv, dtype=np.uint64).copy()
    lib().orc_expand_improved(_p(cv), C.c_uint32(g), C.c_uint32(t_exp), _p(w_left), C.c_uint32(t_exp_right), _p(w_right),
                              C.c_uint32(n_right), C.c_uint32(max_bits_right), C.c_uint32(stopround))
    return cv


def scal_to_mat(cv, w, t_conv):
    out = u64(3, 2, 2, N)
    lib().orc_scal_to_mat(_p(out), _p(np.ascontiguousarray(cv)), _p(np.ascontiguousarray(w)), C.c_uint32(t_conv))
    return out


def regev_to_gsw(cv_v, w, v, t_conv, ell):
    out = u64(3, 3 * ell, 2, N)
    lib().orc_regev_to_gsw(_p(out), _p(np.ascontiguousarray(cv_v)), _p(np.ascontiguousarray(w)), _p(np.ascontiguousarray(v)),
                           C.c_uint32(t_conv), C.c_uint32(ell))
    return out


# ---- stages ----
def stage_expand(p, query, w_left, w_right):
    s = shape_of(p)
    out = u64(s.n_bits, 2, 2, N)
    rc = lib().orc_stage_expand(C.byref(p), _p(query), _p(w_left), _p(w_right), _p(out))
    assert rc == 0
    return out


def stage_convert(p, cv, w, v):
    s = shape_of(p)
    cts = u64(s.dim0, 3, 2, 2, N)
    gsw = u64(max(p.nu2, 1), 3, s.m2, 2, N)
    rc = lib().orc_stage_convert(C.byref(p), _p(cv), _p(w), _p(v), _p(cts), _p(gsw))
    assert rc == 0
    return cts, gsw[: p.nu2]


def stage_first_dim(p, cts, db):
    s = shape_of(p)
    out = u64(s.num_per, 3, 2, N)
    rc = lib().orc_stage_first_dim(C.byref(p), _p(cts), _p(db), _p(out))
    assert rc == 0
    return out


def stage_fold(p, raw_cts, gsw):
    raw = np.ascontiguousarray(raw_cts, dtype=np.uint64).copy()
    out = u64(3, 2, N)
    gsw = np.ascontiguousarray(gsw) if gsw.size else u64(1)
    rc = lib().orc_stage_fold(C.byref(p), _p(raw), _p(gsw), _p(out))
    assert rc == 0
    return out


def stage_rescale(p, final_ct):
    out = u64(3, 2, N)
    rc = lib().orc_stage_rescale(C.byref(p), _p(np.ascontiguousarray(final_ct)), _p(out))
    assert rc == 0
    return out


def answer(p, query, w_left, w_right, w, v, db):
    out = u64(3, 2, N)
    rc = lib().orc_answer(C.byref(p), _p(query), _p(w_left), _p(w_right), _p(w), _p(v), _p(db), _p(out))
    assert rc == 0
    return out


# ---- DB ----
def db_words(p):
    s = shape_of(p)
    return s.dim0 * s.num_per * 4 * N


def gen_db(p, seed):
    db = u64(db_words(p))
    lib().orc_gen_db(C.byref(p), C.c_uint64(seed), _p(db))
    return db


def db_item(p, seed, item):
    out = u64(2, 2, N)
    lib().orc_db_item(C.byref(p), C.c_uint64(seed), C.c_uint64(item), _p(out))
    return out


def encode_item(p, pt):
    """plaintext (n0 x n2 raw coefficients in [0, p_db)) -> its NTT-form encoding: centred lift + to_ntt (src/spiral.cpp:1116-1128)"""
    out = u64(2, 2, 2, N)
    lib().orc_encode_item(C.byref(p), _p(np.ascontiguousarray(pt, dtype=np.uint64)), _p(out))
    return out


def pack_items(pts, coeff_bits):
    """plaintext coefficients (any shape, values < 2^coeff_bits) -> the bit-packed item stream of spiral_gpu_server_load_db_items:
    little-endian bit order, coefficient k at bits [k*coeff_bits, (k+1)*coeff_bits) (read_arbitrary_bits, src/core.cpp:20-30)"""
    v = np.ascontiguousarray(pts, dtype=np.uint64).reshape(-1)
    if coeff_bits in (8, 16, 32, 64):
        return np.ascontiguousarray(v.astype({8: "<u1", 16: "<u2", 32: "<u4", 64: "<u8"}[coeff_bits])).view(np.uint8)
    bits = ((v[:, None] >> np.arange(coeff_bits, dtype=np.uint64)[None, :]) & np.uint64(1)).astype(np.uint8).reshape(-1)
    return np.packbits(bits, bitorder="little")


def ntt_isa() -> str:
    f = lib().orc_ntt_isa
    f.restype = C.c_char_p
    return f().decode()


def set_ntt_simd(on: bool) -> bool:
    """route the transforms through the reference's vector form (AVX2 forward butterflies + closing corrections); returns what is in force"""
    lib().orc_set_ntt_simd.restype = C.c_int
    return bool(lib().orc_set_ntt_simd(C.c_int(1 if on else 0)))


def ntt_forward_scalar(x: np.ndarray) -> np.ndarray:
    out = np.ascontiguousarray(x, dtype=np.uint64).copy()
    flat = out.reshape(-1, NTTP)
    for k in range(flat.shape[0]):
        lib().orc_ntt_forward_scalar(flat[k].ctypes.data_as(_u64p))
    return out


def ntt_inverse_scalar(x: np.ndarray) -> np.ndarray:
    out = np.ascontiguousarray(x, dtype=np.uint64).copy()
    flat = out.reshape(-1, NTTP)
    for k in range(flat.shape[0]):
        lib().orc_ntt_inverse_scalar(flat[k].ctypes.data_as(_u64p))
    return out


def set_threads(n):
    """threads of the OpenMP (`make native`) build; the default build ignores it and returns 1"""
    lib().orc_set_threads.restype = C.c_int
    return lib().orc_set_threads(C.c_int(n))


def fill_db_random(seed, nwords):
    db = u64(nwords)
    lib().orc_fill_db_random(C.c_uint64(seed), _p(db), C.c_size_t(nwords))
    return db


# ---- client ----
class Client:
    def __init__(self, p: Params, seed: int = 1, nonoise: bool = False):
        self.p = p
        self.h = lib().orc_client_new(C.byref(p), C.c_uint64(seed), C.c_int(1 if nonoise else 0))
        if not self.h:
            raise ValueError("unsupported parameter set")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_client_free(C.c_void_p(self.h))
            self.h = None

    def pub_params(self):
        L = lib()
        p = self.p
        wl = u64(max(L.orc_words_w_left(C.byref(p)), 1))
        wr = u64(max(L.orc_words_w_right(C.byref(p)), 1))
        w = u64(L.orc_words_w(C.byref(p)))
        v = u64(L.orc_words_v(C.byref(p)))
        L.orc_client_pub_params(C.c_void_p(self.h), _p(wl), _p(wr), _p(w), _p(v))
        return wl, wr, w, v

    def query(self, idx):
        q = u64(lib().orc_words_query(C.byref(self.p)))
        lib().orc_client_query(C.c_void_p(self.h), C.c_uint64(idx), _p(q))
        return q

    def decode(self, resp):
        out = u64(2, 2, N)
        lib().orc_client_decode(C.c_void_p(self.h), _p(np.ascontiguousarray(resp)), _p(out))
        return out


# ---- SpiralPack (src/testing.cpp) ----
class PackShape(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("dim0", "num_per", "ell", "g", "stopround", "n_left", "n_right", "n_query_cts", "trials")] + [("qprime", C.c_uint64)]


def pack_shape_of(p: Params, out_n: int) -> PackShape:
    s = PackShape()
    if lib().orc_pack_get_shape(C.byref(p), C.c_uint32(out_n), C.byref(s)) != 0:
        raise ValueError("unsupported SpiralPack parameter set")
    return s


def pack_gen_db(p, out_n, seed):
    s = pack_shape_of(p, out_n)
    db = u64(s.trials, s.dim0 * s.num_per * N)
    lib().orc_pack_gen_db(C.byref(p), C.c_uint32(out_n), C.c_uint64(seed), _p(db))
    return db


def pack_gen_db_trial(p, out_n, seed, trial):
    s = pack_shape_of(p, out_n)
    db = u64(s.dim0 * s.num_per * N)
    lib().orc_pack_gen_db_trial(C.byref(p), C.c_uint32(out_n), C.c_uint64(seed), C.c_uint32(trial), _p(db))
    return db


def pack_db_item(p, out_n, seed, item):
    out = u64(out_n, out_n, N)
    lib().orc_pack_db_item(C.byref(p), C.c_uint32(out_n), C.c_uint64(seed), C.c_uint64(item), _p(out))
    return out


def pack_answer(p, out_n, query, w_left, w_right, v, v_w, db):
    resp = u64(out_n + 1, out_n, N)
    fin = u64(out_n + 1, out_n, 2, N)
    rc = lib().orc_pack_answer(C.byref(p), C.c_uint32(out_n), _p(query), _p(w_left), _p(w_right), _p(v), _p(v_w), _p(np.ascontiguousarray(db)), _p(resp), _p(fin))
    assert rc == 0
    return resp, fin


def reorient_dim1(cts, dim0, idx_factor):
    out = u64(N, dim0, 2)
    lib().orc_reorient_dim1(_p(out), _p(np.ascontiguousarray(cts)), C.c_size_t(dim0), C.c_size_t(idx_factor))
    return out


def sweep_dim1(db, reoriented, dim0, num_per):
    out = u64(num_per, 2, 2, N)
    lib().orc_sweep_dim1(_p(out), _p(np.ascontiguousarray(db)), _p(reoriented), C.c_size_t(dim0), C.c_size_t(num_per))
    return out


def regev_to_simple_gsw(cv, v, t_conv, ell, nu2):
    out = u64(nu2, 2, 2 * ell, 2, N)
    lib().orc_regev_to_simple_gsw(_p(out), _p(np.ascontiguousarray(cv)), _p(np.ascontiguousarray(v)), C.c_uint32(t_conv), C.c_uint32(ell), C.c_uint32(nu2))
    return out


def pack_fold_neg(gsw, ell, nu2):
    out = np.zeros_like(gsw)
    lib().orc_pack_fold_neg(_p(out), _p(np.ascontiguousarray(gsw)), C.c_uint32(ell), C.c_uint32(nu2))
    return out


def fold_dim1(raw_cts, folding, folding_neg, ell, nu2):
    x = np.ascontiguousarray(raw_cts, dtype=np.uint64).copy()
    lib().orc_fold_dim1(_p(x), C.c_size_t(x.shape[0]), _p(np.ascontiguousarray(folding)), _p(np.ascontiguousarray(folding_neg)), C.c_uint32(ell), C.c_uint32(nu2))
    return x[0]


def pack(v_ct, v_w, out_n, t_conv):
    out = u64(out_n + 1, out_n, 2, N)
    lib().orc_pack(_p(out), C.c_uint32(out_n), C.c_uint32(t_conv), _p(np.ascontiguousarray(v_ct)), _p(np.ascontiguousarray(v_w)))
    return out


class PackClient:
    def __init__(self, p: Params, out_n: int, seed: int = 1, nonoise: bool = False):
        self.p, self.out_n = p, out_n
        self.h = lib().orc_pack_client_new(C.byref(p), C.c_uint32(out_n), C.c_uint64(seed), C.c_int(1 if nonoise else 0))
        if not self.h:
            raise ValueError("unsupported SpiralPack parameter set")

    def __del__(self):
        if getattr(self, "h", None):
            lib().orc_pack_client_free(C.c_void_p(self.h))
            self.h = None

    def pub_params(self):
        L, p = lib(), self.p
        s = pack_shape_of(p, self.out_n)
        wl = u64(max(s.n_left * 2 * p.t_exp * NTTP, 1))
        wr = u64(max(s.n_right * 2 * p.t_exp_right * NTTP, 1))
        v = u64(L.orc_pack_words_v(C.byref(p)))
        vw = u64(L.orc_pack_words_vw(C.byref(p), C.c_uint32(self.out_n)))
        L.orc_pack_client_pub_params(C.c_void_p(self.h), _p(wl), _p(wr), _p(v), _p(vw))
        return wl, wr, v, vw

    def query(self, idx):
        q = u64(lib().orc_pack_words_query(C.byref(self.p), C.c_uint32(self.out_n)))
        lib().orc_pack_client_query(C.c_void_p(self.h), C.c_uint64(idx), _p(q))
        return q

    def decode(self, resp):
        out = u64(self.out_n, self.out_n, N)
        lib().orc_pack_client_decode(C.c_void_p(self.h), _p(np.ascontiguousarray(resp)), _p(out))
        return out

"""SpiralPack / SpiralStreamPack (`--high-rate`, reference src/testing.cpp): host mirror of the function seams and
the resident server handle, each a direct call through the C ABI."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import U64P, PackShape, Params, check, lib

N = 2048
PACK_STAGE_NAMES = ["expansion_us", "conversion_us", "first_dim_us", "folding_us", "packing_us", "sweep_kernels_us", "total_us", "reserved"]


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(U64P)


def _c(a) -> np.ndarray:
    return np.ascontiguousarray(a, dtype=np.uint64)


def get_pack_shape(p: Params, out_n: int) -> PackShape:
    s = PackShape()
    check(lib().spiral_gpu_pack_get_shape(C.byref(p), out_n, C.byref(s)))
    return s


def pack(out_n, m_conv, v_ct, v_W) -> np.ndarray:
    """pack (include/testing.h:36): out_n^2 raw 2x1 cts + out_n key matrices -> (out_n+1) x out_n NTT ciphertext"""
    out = np.zeros((out_n + 1, out_n, 2, N), dtype=np.uint64)
    check(lib().spiral_gpu_pack(_p(out), out_n, m_conv, _p(_c(v_ct)), _p(_c(v_W))))
    return out


def fastMultiplyQueryByDatabaseDim1(db, v_firstdim, dim0, num_per) -> np.ndarray:
    out = np.zeros((num_per, 2, 2, N), dtype=np.uint64)
    check(lib().spiral_gpu_fast_multiply_query_by_database_dim1(_p(out), _p(_c(db)), _p(_c(v_firstdim)), dim0, num_per))
    return out


class PackServer:
    def __init__(self, params: Params, out_n: int, device: int = 0, trial0: int = 0, trial1: int = 0):
        """trial0, trial1: this server's share [trial0, trial1) of the out_n^2 trials (N GPUs); 0, 0 = all of them"""
        self.params, self.out_n = params, out_n
        self.shape = get_pack_shape(params, out_n)
        h = C.c_void_p()
        check(lib().spiral_gpu_pack_server_create_sharded(C.byref(params), out_n, device, trial0, trial1, C.byref(h)))
        self.h = h
        self.trial0, self.trial1 = (trial0, trial1) if trial1 else (0, self.shape.trials)

    def set_stream(self, hip_stream: int):
        check(lib().spiral_gpu_pack_server_set_stream(self.h, C.c_void_p(hip_stream)))

    def fold_trials(self, query, folded_ptr: int):
        """expansion + conversion + this server's sweeps and folding; the folded ciphertexts ([n_local][2][N] raw u64) are written to
        the device buffer at folded_ptr (asynchronous on the server's stream)"""
        check(lib().spiral_gpu_pack_server_fold_trials(self.h, _p(_c(query)), C.c_void_p(folded_ptr)))

    def stage_us(self) -> dict:
        """stage times of the last answer / fold_trials (synchronises the stream)"""
        us = (C.c_double * 8)()
        check(lib().spiral_gpu_pack_server_stage_us(self.h, us))
        return dict(zip(PACK_STAGE_NAMES, list(us)))

    def pack_gathered(self, gathered_ptr: int, want_packed: bool = False):
        """pack + modulus switch of all out_n^2 folded ciphertexts (device buffer in trial order): (response, packed or None)"""
        n = self.out_n
        resp = np.zeros((n + 1, n, N), dtype=np.uint64)
        packed = np.zeros((n + 1, n, 2, N), dtype=np.uint64) if want_packed else None
        check(lib().spiral_gpu_pack_server_pack_gathered(self.h, C.c_void_p(gathered_ptr), _p(resp), _p(packed) if want_packed else None))
        return resp, packed

    def close(self):
        if getattr(self, "h", None):
            lib().spiral_gpu_pack_server_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def gen_db(self, seed: int):
        check(lib().spiral_gpu_pack_server_gen_db(self.h, seed))

    def load_db(self, trial: int, db):
        check(lib().spiral_gpu_pack_server_load_db(self.h, trial, _p(_c(db))))

    def load_db_items(self, trial: int, items, coeff_bits: int, first_item: int = 0, n_items=None):
        """raw ingest of one trial: bit-packed plaintext coefficients (coeff_bits each), 2048 per item"""
        items = np.ascontiguousarray(items)
        if n_items is None:
            n_items = items.nbytes * 8 // (N * coeff_bits)
        check(lib().spiral_gpu_pack_server_load_db_items(self.h, trial, items.ctypes.data_as(C.c_void_p), coeff_bits, first_item, n_items))

    def read_acc(self, trial: int) -> np.ndarray:
        """first-dimension accumulators of one trial of the last answer: [num_per][2][2][N] NTT form"""
        out = np.zeros((self.shape.num_per, 2, 2, N), dtype=np.uint64)
        check(lib().spiral_gpu_pack_server_read_acc(self.h, trial, _p(out)))
        return out

    def read_response_wire(self) -> np.ndarray:
        """the last answer's response in its wire form (bit-packed on the device): bytes"""
        n = lib().spiral_gpu_response_wire_bytes(C.byref(self.params), self.out_n)
        out = np.zeros(n, dtype=np.uint8)
        check(lib().spiral_gpu_pack_server_read_response_wire(self.h, out.ctypes.data_as(C.c_void_p), n))
        return out

    def fill_db_random(self, seed: int):
        check(lib().spiral_gpu_pack_server_fill_db_random(self.h, seed))

    def set_pub_params(self, w_left, w_right, v, v_w):
        check(lib().spiral_gpu_pack_server_set_pub_params(self.h, _p(w_left), _p(w_right), _p(v), _p(v_w)))

    def answer(self, query, want_packed: bool = True):
        n = self.out_n
        resp = np.zeros((n + 1, n, N), dtype=np.uint64)
        packed = np.zeros((n + 1, n, 2, N), dtype=np.uint64) if want_packed else None
        us = (C.c_double * 8)()
        check(lib().spiral_gpu_pack_server_answer(self.h, _p(_c(query)), _p(resp), _p(packed) if want_packed else None, us))
        return resp, packed, dict(zip(PACK_STAGE_NAMES, list(us)))

    def sweep_bytes(self) -> int:
        return int(lib().spiral_gpu_pack_server_sweep_bytes(self.h))

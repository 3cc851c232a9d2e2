"""Resident server handle: the server half of do_test (src/spiral.cpp:2337-2406, 1584-1629) with the
database, public parameters and intermediates kept in HBM."""
from __future__ import annotations

import ctypes as C

import numpy as np

from ._lib import U64P, Params, Shape, check, lib

N = 2048

BUF_EXPANDED, BUF_CTS, BUF_GSW, BUF_ACC, BUF_RAW, BUF_FINAL, BUF_RESPONSE = range(7)
DB_PACKED, DB_LIMBS = 0, 1  # spiral_gpu_db_format
STAGE_NAMES = ["expansion_us", "conversion_us", "first_dim_us", "folding_us", "response_us", "sweep_kernel_us", "total_us", "scaltomat_us"]


def _p(a: np.ndarray):
    assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(U64P)


def first_dim_batch(servers):
    """one pass over the database for the (converted) queries of up to eight servers sharing one image; see include/spiral_gpu.h"""
    arr = (C.c_void_p * len(servers))(*[s.h for s in servers])
    check(lib().spiral_gpu_server_first_dim_batch(arr, len(servers)))


def time_sweep_batch(servers, iters: int = 20) -> float:
    """average ms of first_dim_batch's sweep launch alone (HIP events on servers[0]'s stream)"""
    arr = (C.c_void_p * len(servers))(*[s.h for s in servers])
    ms = C.c_float()
    check(lib().spiral_gpu_server_time_sweep_batch(arr, len(servers), iters, C.byref(ms)))
    return ms.value


def run_query_batch(servers):
    """the whole answer for the queries of up to eight servers sharing one image, every launch carrying all of them; see include/spiral_gpu.h"""
    arr = (C.c_void_p * len(servers))(*[s.h for s in servers])
    check(lib().spiral_gpu_server_run_query_batch(arr, len(servers)))


class Server:
    def __init__(self, params: Params, device: int = 0, j_begin: int = 0, j_end: int = 0, share_db_of: "Server | None" = None):
        """share_db_of: make this server a query lane of that one -- same parameters, device and shard, sweeping ITS database
        image (no second image is ever allocated)"""
        self.params = params
        self.shape = Shape()
        check(lib().spiral_gpu_get_shape(C.byref(params), C.byref(self.shape)))
        h = C.c_void_p()
        if share_db_of is not None:
            check(lib().spiral_gpu_server_create_lane(share_db_of.h, C.byref(h)))
            self._db_owner = share_db_of  # keeps the owner alive
        else:
            check(lib().spiral_gpu_server_create(C.byref(params), device, j_begin, j_end, C.byref(h)))
        self.h = h
        self.dim0_shard = (j_end - j_begin) if (j_begin or j_end) else self.shape.dim0

    def close(self):
        if getattr(self, "h", None):
            lib().spiral_gpu_server_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs ----
    def set_stream(self, stream_ptr: int):
        check(lib().spiral_gpu_server_set_stream(self.h, C.c_void_p(stream_ptr)))

    def load_db(self, database: np.ndarray):
        check(lib().spiral_gpu_server_load_db(self.h, _p(np.ascontiguousarray(database, dtype=np.uint64))))

    def gen_db(self, seed: int):
        check(lib().spiral_gpu_server_gen_db(self.h, seed))

    def load_db_items(self, items: np.ndarray, coeff_bits: int, first_item: int = 0, n_items: int | None = None):
        """raw ingest: bit-packed plaintext coefficients (uint8 stream, coeff_bits each) or raw u64 MatPoly words (coeff_bits = 64)"""
        items = np.ascontiguousarray(items)
        if n_items is None:
            n_items = items.nbytes * 8 // (4 * N * coeff_bits)
        check(lib().spiral_gpu_server_load_db_items(self.h, items.ctypes.data_as(C.c_void_p), coeff_bits, first_item, n_items))

    def read_db_item(self, item: int) -> np.ndarray:
        out = np.zeros((2, 2, 2, N), dtype=np.uint64)
        check(lib().spiral_gpu_server_read_db_item(self.h, item, _p(out)))
        return out

    def read_db_slots(self, z_begin: int, nz: int = 1) -> np.ndarray:
        """load_db's layout restricted to this server's j-range: [nz][num_per][n2][j][n0] packed words"""
        out = np.zeros((nz, self.shape.num_per, 2, self.dim0_shard, 2), dtype=np.uint64)
        check(lib().spiral_gpu_server_read_db_slots(self.h, z_begin, nz, _p(out)))
        return out

    def read_db_columns(self, ii_begin: int, n_ii: int = 1) -> np.ndarray:
        """all 2048 slots of the plaintext columns ii_begin .. ii_begin + n_ii - 1: [2048][n_ii][n2][j][n0] packed words"""
        out = np.zeros((N, n_ii, 2, self.dim0_shard, 2), dtype=np.uint64)
        check(lib().spiral_gpu_server_read_db_columns(self.h, ii_begin, n_ii, _p(out)))
        return out

    def fill_db_random(self, seed: int):
        check(lib().spiral_gpu_server_fill_db_random(self.h, seed))

    def run_query_instances(self, instances, responses_ptr: int, finals_ptr: int = 0, pre: bool = True):
        """this server's query against every database instance in `instances` (servers holding the images of one item's `factor` databases): expansion +
        conversion once (pre), then sweep + folding + switch per instance; responses_ptr / finals_ptr: device memory, 6 x 2048 words per instance"""
        arr = (C.c_void_p * len(instances))(*[s.h for s in instances])
        check(lib().spiral_gpu_server_run_query_instances(self.h, arr, len(instances), 1 if pre else 0, C.c_void_p(responses_ptr), C.c_void_p(finals_ptr or None)))

    def answer_instances(self, instances, query):
        """host-buffer form of run_query_instances: (responses [n][3][2][N], folded ciphertexts [n][3][2][N], device us of the item query)"""
        arr = (C.c_void_p * len(instances))(*[s.h for s in instances])
        n = len(instances)
        resp, fin, us = np.zeros((n, 3, 2, N), dtype=np.uint64), np.zeros((n, 3, 2, N), dtype=np.uint64), C.c_double()
        check(lib().spiral_gpu_server_answer_instances(self.h, arr, n, _p(np.ascontiguousarray(query, dtype=np.uint64)), _p(resp), _p(fin), C.byref(us)))
        return resp, fin, us.value

    def set_db_format(self, fmt: int):
        """convert this server's database image in place: DB_PACKED (vector-ALU sweep) <-> DB_LIMBS (matrix-core sweep); see include/spiral_gpu.h"""
        check(lib().spiral_gpu_server_set_db_format(self.h, fmt))

    def db_format(self) -> int:
        return lib().spiral_gpu_server_db_format(self.h)

    def db_device_bytes(self) -> int:
        """device bytes the holder of this server's image keeps for database images"""
        return lib().spiral_gpu_server_db_device_bytes(self.h)

    def share_db(self, owner: "Server"):
        """sweep `owner`'s database image instead of an own copy (a second query lane on one database)"""
        check(lib().spiral_gpu_server_share_db(self.h, owner.h))
        self._db_owner = owner  # keeps the owner alive

    def set_pub_params(self, w_left, w_right, w, v):
        check(lib().spiral_gpu_server_set_pub_params(self.h, _p(w_left), _p(w_right), _p(w), _p(v)))

    def set_query(self, query):
        check(lib().spiral_gpu_server_set_query(self.h, _p(np.ascontiguousarray(query, dtype=np.uint64))))

    # ---- stages ----
    def expand(self):
        check(lib().spiral_gpu_server_expand(self.h))

    def convert(self):
        check(lib().spiral_gpu_server_convert(self.h))

    def first_dim(self):
        check(lib().spiral_gpu_server_first_dim(self.h))

    def lift(self, reduce_first: bool = False):
        check(lib().spiral_gpu_server_lift(self.h, 1 if reduce_first else 0))

    def fold(self):
        check(lib().spiral_gpu_server_fold(self.h))

    def finish(self):
        check(lib().spiral_gpu_server_finish(self.h))

    def sync(self):
        check(lib().spiral_gpu_server_sync(self.h))

    def use_graphs(self, on: bool = True):
        check(lib().spiral_gpu_server_use_graphs(self.h, 1 if on else 0))

    def set_overlap(self, on=2):
        """0: one stream; 2: the split schedule -- the whole GSW side of the query (the odd tree of the expansion + the conversion) as its own
        launch sequence on the side stream beside the even tree + ScalToMat + sweep (same results, different schedule)"""
        check(lib().spiral_gpu_server_set_overlap(self.h, int(on)))

    def run_pre(self):
        """expand + convert (one hipGraph replay when graphs are on)"""
        check(lib().spiral_gpu_server_run_pre(self.h))

    def run_query(self):
        """run_pre + first_dim + run_post as one group (one hipGraph replay when graphs are on); single GPU only"""
        check(lib().spiral_gpu_server_run_query(self.h))

    def run_pre_sweep(self):
        """run_pre + first_dim as one group: what a rank does before the collective of a sharded answer"""
        check(lib().spiral_gpu_server_run_pre_sweep(self.h))

    def run_post(self, reduce_first: bool = False):
        """lift + fold + finish"""
        check(lib().spiral_gpu_server_run_post(self.h, 1 if reduce_first else 0))

    def set_fold_ranks(self, n_ranks: int):
        check(lib().spiral_gpu_server_set_fold_ranks(self.h, n_ranks))

    def fold_local(self, acc_chunk_ptr: int, out_ct_ptr: int):
        """lift this rank's reduce-scattered chunk and run the local folding rounds -> one raw ct at out_ct_ptr"""
        check(lib().spiral_gpu_server_fold_local(self.h, C.c_void_p(acc_chunk_ptr), C.c_void_p(out_ct_ptr)))

    def fold_root(self, gathered_ptr: int):
        """last log2(G) folding rounds + response switch on the G gathered cts (rank order)"""
        check(lib().spiral_gpu_server_fold_root(self.h, C.c_void_p(gathered_ptr)))

    def set_expand_shard(self, rank: int, n_ranks: int):
        """expand only what rank `rank` of an n_ranks-GPU answer needs (own first-dimension block, every n_ranks-th GSW bit)"""
        check(lib().spiral_gpu_server_set_expand_shard(self.h, rank, n_ranks))

    def gsw_bits_words(self) -> int:
        return int(lib().spiral_gpu_server_gsw_bits_words(self.h))

    def gsw_bits_pack(self, block_ptr: int):
        check(lib().spiral_gpu_server_gsw_bits_pack(self.h, C.c_void_p(block_ptr)))

    def gsw_bits_unpack(self, gathered_ptr: int):
        check(lib().spiral_gpu_server_gsw_bits_unpack(self.h, C.c_void_p(gathered_ptr)))

    def run_expand_pack(self, block_ptr: int):
        """sharded expansion + pack of this rank's GSW bits (one hipGraph replay when graphs are on)"""
        check(lib().spiral_gpu_server_run_expand_pack(self.h, C.c_void_p(block_ptr)))

    def run_unpack_convert_sweep(self, gathered_ptr: int):
        """unpack of the all-gathered GSW bits + convert + first_dim (one hipGraph replay when graphs are on)"""
        check(lib().spiral_gpu_server_run_unpack_convert_sweep(self.h, C.c_void_p(gathered_ptr)))

    def run_scal2mat_sweep(self):
        """ScalToMat + first_dim: the database-dependent part needs no GSW bit, so it can run under their all-gather"""
        check(lib().spiral_gpu_server_run_scal2mat_sweep(self.h))

    def set_sweep_stages(self, n_stages: int):
        """pipelined sweep: accumulators laid out [stage][rank][ct], so each stage's 1/n of the buffer can be reduce-scattered while the next sweeps"""
        check(lib().spiral_gpu_server_set_sweep_stages(self.h, n_stages))

    def max_sweep_stages(self) -> int:
        return int(lib().spiral_gpu_server_max_sweep_stages(self.h))

    def first_dim_stage(self, stage: int):
        check(lib().spiral_gpu_server_first_dim_stage(self.h, stage))

    def run_scal2mat(self):
        """ScalToMat alone (the pipelined schedule issues the sweep stage by stage after it)"""
        check(lib().spiral_gpu_server_run_scal2mat(self.h))

    def run_unpack_gsw(self, gathered_ptr: int):
        """unpack of the all-gathered GSW bits + Regev->GSW conversion (fold keys)"""
        check(lib().spiral_gpu_server_run_unpack_gsw(self.h, C.c_void_p(gathered_ptr)))

    def acc(self):
        nbytes = C.c_size_t()
        ptr = lib().spiral_gpu_server_acc(self.h, C.byref(nbytes))
        return ptr, nbytes.value

    def set_acc(self, device_ptr: int):
        check(lib().spiral_gpu_server_set_acc(self.h, C.c_void_p(device_ptr)))

    def answer(self, query):
        """process_crtd_query on one query -> (final raw ct n1 x n2, response, stage times in us)"""
        fin = np.zeros((3, 2, N), dtype=np.uint64)
        resp = np.zeros((3, 2, N), dtype=np.uint64)
        us = (C.c_double * 8)()
        check(lib().spiral_gpu_server_answer(self.h, _p(np.ascontiguousarray(query, dtype=np.uint64)), _p(fin), _p(resp), us))
        return fin, resp, dict(zip(STAGE_NAMES, list(us)))

    def answer_resident(self):
        us = (C.c_double * 8)()
        check(lib().spiral_gpu_server_answer_resident(self.h, us))
        return dict(zip(STAGE_NAMES, list(us)))

    # ---- introspection ----
    def keep_cts(self, on: bool = True):
        check(lib().spiral_gpu_server_keep_cts(self.h, 1 if on else 0))

    def read_response_wire(self) -> np.ndarray:
        """the last answer's response in its wire form (bit-packed on the device): bytes"""
        n = lib().spiral_gpu_response_wire_bytes(C.byref(self.params), 2)
        out = np.zeros(n, dtype=np.uint8)
        check(lib().spiral_gpu_server_read_response_wire(self.h, out.ctypes.data_as(C.c_void_p), n))
        return out

    def read(self, which: int) -> np.ndarray:
        words = lib().spiral_gpu_server_buffer_words(self.h, which)
        out = np.zeros(words, dtype=np.uint64)
        check(lib().spiral_gpu_server_read(self.h, which, _p(out)))
        s, p = self.shape, self.params
        shapes = {
            BUF_EXPANDED: (s.n_bits, 2, 2, N),
            BUF_CTS: (-1, 3, 2, 2, N),
            BUF_GSW: (p.nu2, 3, s.m2, 2, N),
            BUF_ACC: (s.num_per, 3, 2, 2, N),
            BUF_RAW: (s.num_per, 3, 2, N),
            BUF_FINAL: (3, 2, N),
            BUF_RESPONSE: (3, 2, N),
        }
        return out.reshape(shapes[which])

    def write_raw(self, raw_cts):
        check(lib().spiral_gpu_server_write_raw(self.h, _p(np.ascontiguousarray(raw_cts, dtype=np.uint64))))

    def time_sweep(self, iters: int = 20) -> float:
        ms = C.c_float()
        check(lib().spiral_gpu_server_time_sweep(self.h, iters, C.byref(ms)))
        return ms.value

    def sweep_bytes(self) -> int:
        return int(lib().spiral_gpu_server_sweep_bytes(self.h))

    def sweep_device_bytes(self) -> int:
        """bytes one sweep launch has to move on this device (packed database + query records + accumulators)"""
        return int(lib().spiral_gpu_server_sweep_device_bytes(self.h))

"""ctypes binding of libspiral_gpu.so (the C ABI declared in include/spiral_gpu.h).

The library is the product; this module only loads it and declares prototypes.  It fails loudly when
the shared object is missing -- there is no CPU fallback anywhere in this package.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
# SPIRAL_LIB=<path>: load another build of the library (tuning A/B runs, tools/build_variants.sh: -DSPIRAL_TUNING builds of the same sources);
# every process of a multi-rank run inherits it.  Whatever is loaded must export every declared entry point.
LIB_PATH = os.environ.get("SPIRAL_LIB") or os.path.join(HERE, "libspiral_gpu.so")
CSRC = os.path.join(HERE, "csrc")


class Params(C.Structure):
    """spiral_gpu_params: the reference's -D scheme parameters (include/values.h:78-93) + argv[1..2]"""

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


class PackShape(C.Structure):
    _fields_ = [(n, C.c_uint32) for n in ("dim0", "num_per", "ell", "g", "stopround", "n_left", "n_right", "n_query_cts", "trials")] + [
        ("qprime", C.c_uint64)
    ]


U64P = C.POINTER(C.c_uint64)

# name -> (restype, argtypes); every symbol include/spiral_gpu.h declares
PROTOTYPES = {
    "spiral_gpu_abi_version": (C.c_int, []),
    "spiral_gpu_last_error": (C.c_char_p, []),
    "spiral_gpu_device_count": (C.c_int, []),
    "spiral_gpu_get_shape": (C.c_int, [C.POINTER(Params), C.POINTER(Shape)]),
    "spiral_gpu_set_option": (C.c_int, [C.c_char_p, C.c_int64]),
    "spiral_gpu_get_option": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64)]),
    "spiral_gpu_get_tables": (C.c_int, [U64P]),
    "spiral_gpu_ntt_forward": (C.c_int, [U64P, C.c_size_t]),
    "spiral_gpu_ntt_inverse": (C.c_int, [U64P, C.c_size_t]),
    "spiral_gpu_to_ntt": (C.c_int, [U64P, U64P, C.c_size_t, C.c_int]),
    "spiral_gpu_from_ntt": (C.c_int, [U64P, U64P, C.c_size_t]),
    "spiral_gpu_time_ntt": (C.c_int, [C.c_size_t, C.c_int, C.POINTER(C.c_float), C.POINTER(C.c_float)]),
    "spiral_gpu_time_ntt_digits": (C.c_int, [C.c_size_t, C.c_uint32, C.c_int, C.POINTER(C.c_float)]),
    "spiral_gpu_multiply": (C.c_int, [U64P, U64P, U64P, C.c_size_t, C.c_size_t, C.c_size_t]),
    "spiral_gpu_add": (C.c_int, [U64P, U64P, U64P, C.c_size_t]),
    "spiral_gpu_mul_by_const": (C.c_int, [U64P, U64P, U64P, C.c_size_t]),
    "spiral_gpu_automorph": (C.c_int, [U64P, U64P, C.c_size_t, C.c_uint64]),
    "spiral_gpu_invert": (C.c_int, [U64P, U64P, C.c_size_t]),
    "spiral_gpu_gadget_invert": (C.c_int, [U64P, U64P, C.c_size_t, C.c_size_t, C.c_size_t]),
    "spiral_gpu_get_rescaled": (C.c_int, [U64P, U64P, C.c_size_t, C.c_uint64, C.c_uint64]),
    "spiral_gpu_multiply_query_by_database": (C.c_int, [U64P, U64P, U64P, C.c_size_t, C.c_size_t]),
    "spiral_gpu_multiply_queries_by_database": (C.c_int, [U64P, U64P, C.c_size_t, U64P, C.c_size_t, C.c_size_t]),
    "spiral_gpu_split_and_crt": (C.c_int, [U64P, U64P, C.c_size_t, C.c_uint32]),
    "spiral_gpu_fold_one_further_dimension": (C.c_int, [U64P, C.c_size_t, U64P, U64P, C.c_uint32]),
    "spiral_gpu_expand_improved": (C.c_int, [U64P, C.c_uint32, C.c_uint32, U64P, C.c_uint32, U64P, C.c_uint32, C.c_uint32, C.c_uint32]),
    "spiral_gpu_scal_to_mat": (C.c_int, [U64P, U64P, U64P, C.c_uint32]),
    "spiral_gpu_regev_to_gsw": (C.c_int, [U64P, U64P, U64P, U64P, C.c_uint32, C.c_uint32]),
    "spiral_gpu_server_create": (C.c_int, [C.POINTER(Params), C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "spiral_gpu_server_destroy": (None, [C.c_void_p]),
    "spiral_gpu_server_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spiral_gpu_server_get_stream": (C.c_void_p, [C.c_void_p]),
    "spiral_gpu_server_load_db": (C.c_int, [C.c_void_p, U64P]),
    "spiral_gpu_server_gen_db": (C.c_int, [C.c_void_p, C.c_uint64]),
    "spiral_gpu_server_fill_db_random": (C.c_int, [C.c_void_p, C.c_uint64]),
    "spiral_gpu_server_set_db_format": (C.c_int, [C.c_void_p, C.c_int]),
    "spiral_gpu_server_db_format": (C.c_int, [C.c_void_p]),
    "spiral_gpu_server_db_device_bytes": (C.c_uint64, [C.c_void_p]),
    "spiral_gpu_server_share_db": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spiral_gpu_server_create_lane": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p)]),
    "spiral_gpu_server_first_dim_batch": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32]),
    "spiral_gpu_server_run_query_batch": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32]),
    "spiral_gpu_server_run_query_instances": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, C.c_int, C.c_void_p, C.c_void_p]),
    "spiral_gpu_server_answer_instances": (C.c_int, [C.c_void_p, C.POINTER(C.c_void_p), C.c_uint32, U64P, U64P, U64P, C.POINTER(C.c_double)]),
    "spiral_gpu_response_wire_bytes": (C.c_size_t, [C.POINTER(Params), C.c_uint32]),
    "spiral_gpu_response_from_wire": (C.c_int, [C.POINTER(Params), C.c_uint32, C.c_void_p, U64P]),
    "spiral_gpu_server_read_response_wire": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "spiral_gpu_pack_server_read_response_wire": (C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    "spiral_gpu_server_load_db_items": (C.c_int, [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]),
    "spiral_gpu_server_read_db_item": (C.c_int, [C.c_void_p, C.c_uint64, U64P]),
    "spiral_gpu_server_read_db_slots": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, U64P]),
    "spiral_gpu_server_read_db_columns": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32, U64P]),
    "spiral_gpu_server_set_pub_params": (C.c_int, [C.c_void_p, U64P, U64P, U64P, U64P]),
    "spiral_gpu_server_set_query": (C.c_int, [C.c_void_p, U64P]),
    "spiral_gpu_server_expand": (C.c_int, [C.c_void_p]),
    "spiral_gpu_server_convert": (C.c_int, [C.c_void_p]),
    "spiral_gpu_server_first_dim": (C.c_int, [C.c_void_p]),
    "spiral_gpu_server_lift": (C.c_int, [C.c_void_p, C.c_int]),
    "spiral_gpu_server_fold": (C.c_int, [C.c_void_p]),
    "spiral_gpu_server_finish": (C.c_int, [C.c_void_p]),
    "spiral_gpu_server_sync": (C.c_int, [C.c_void_p]),
    "spiral_gpu_server_use_graphs": (C.c_int, [C.c_void_p, C.c_int]),
    "spiral_gpu_server_set_overlap": (C.c_int, [C.c_void_p, C.c_int]),
    "spiral_gpu_server_run_query": (C.c_int, [C.c_void_p]),
    "spiral_gpu_server_run_pre_sweep": (C.c_int, [C.c_void_p]),
    "spiral_gpu_server_run_pre": (C.c_int, [C.c_void_p]),
    "spiral_gpu_server_run_post": (C.c_int, [C.c_void_p, C.c_int]),
    "spiral_gpu_server_set_fold_ranks": (C.c_int, [C.c_void_p, C.c_uint32]),
    "spiral_gpu_server_fold_local": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p]),
    "spiral_gpu_server_fold_root": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spiral_gpu_server_set_expand_shard": (C.c_int, [C.c_void_p, C.c_uint32, C.c_uint32]),
    "spiral_gpu_server_gsw_bits_words": (C.c_size_t, [C.c_void_p]),
    "spiral_gpu_server_gsw_bits_pack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spiral_gpu_server_gsw_bits_unpack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spiral_gpu_server_run_expand_pack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spiral_gpu_server_run_unpack_convert_sweep": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spiral_gpu_server_run_scal2mat_sweep": (C.c_int, [C.c_void_p]),
    "spiral_gpu_server_run_unpack_gsw": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spiral_gpu_server_set_sweep_stages": (C.c_int, [C.c_void_p, C.c_uint32]),
    "spiral_gpu_server_max_sweep_stages": (C.c_uint32, [C.c_void_p]),
    "spiral_gpu_server_first_dim_stage": (C.c_int, [C.c_void_p, C.c_uint32]),
    "spiral_gpu_server_run_scal2mat": (C.c_int, [C.c_void_p]),
    "spiral_gpu_server_acc": (C.c_void_p, [C.c_void_p, C.POINTER(C.c_size_t)]),
    "spiral_gpu_server_set_acc": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spiral_gpu_server_answer": (C.c_int, [C.c_void_p, U64P, U64P, U64P, C.POINTER(C.c_double)]),
    "spiral_gpu_server_answer_resident": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "spiral_gpu_server_keep_cts": (C.c_int, [C.c_void_p, C.c_int]),
    "spiral_gpu_server_buffer_words": (C.c_size_t, [C.c_void_p, C.c_int]),
    "spiral_gpu_server_read": (C.c_int, [C.c_void_p, C.c_int, U64P]),
    "spiral_gpu_server_write_raw": (C.c_int, [C.c_void_p, U64P]),
    "spiral_gpu_server_time_sweep": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "spiral_gpu_server_time_sweep_batch": (C.c_int, [C.POINTER(C.c_void_p), C.c_uint32, C.c_int, C.POINTER(C.c_float)]),
    "spiral_gpu_server_sweep_bytes": (C.c_uint64, [C.c_void_p]),
    "spiral_gpu_server_sweep_device_bytes": (C.c_uint64, [C.c_void_p]),
    "spiral_gpu_pack_get_shape": (C.c_int, [C.POINTER(Params), C.c_uint32, C.POINTER(PackShape)]),
    "spiral_gpu_pack": (C.c_int, [U64P, C.c_uint32, C.c_uint32, U64P, U64P]),
    "spiral_gpu_fast_multiply_query_by_database_dim1": (C.c_int, [U64P, U64P, U64P, C.c_size_t, C.c_size_t]),
    "spiral_gpu_pack_server_create": (C.c_int, [C.POINTER(Params), C.c_uint32, C.c_int, C.POINTER(C.c_void_p)]),
    "spiral_gpu_pack_server_create_sharded": (C.c_int, [C.POINTER(Params), C.c_uint32, C.c_int, C.c_uint32, C.c_uint32, C.POINTER(C.c_void_p)]),
    "spiral_gpu_pack_server_set_stream": (C.c_int, [C.c_void_p, C.c_void_p]),
    "spiral_gpu_pack_server_stage_us": (C.c_int, [C.c_void_p, C.POINTER(C.c_double)]),
    "spiral_gpu_pack_server_fold_trials": (C.c_int, [C.c_void_p, U64P, C.c_void_p]),
    "spiral_gpu_pack_server_pack_gathered": (C.c_int, [C.c_void_p, C.c_void_p, U64P, U64P]),
    "spiral_gpu_pack_server_destroy": (None, [C.c_void_p]),
    "spiral_gpu_pack_server_gen_db": (C.c_int, [C.c_void_p, C.c_uint64]),
    "spiral_gpu_pack_server_load_db": (C.c_int, [C.c_void_p, C.c_uint32, U64P]),
    "spiral_gpu_pack_server_fill_db_random": (C.c_int, [C.c_void_p, C.c_uint64]),
    "spiral_gpu_pack_server_load_db_items": (C.c_int, [C.c_void_p, C.c_uint32, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64]),
    "spiral_gpu_pack_server_set_pub_params": (C.c_int, [C.c_void_p, U64P, U64P, U64P, U64P]),
    "spiral_gpu_pack_server_answer": (C.c_int, [C.c_void_p, U64P, U64P, U64P, C.POINTER(C.c_double)]),
    "spiral_gpu_pack_server_read_acc": (C.c_int, [C.c_void_p, C.c_uint32, U64P]),
    "spiral_gpu_pack_server_sweep_bytes": (C.c_uint64, [C.c_void_p]),
}


def build(force: bool = False) -> str:
    """hipcc-compile the HIP extension for gfx950 (works without a GPU)."""
    args = ["make", "-C", CSRC, "-s", "-j4"]
    if force:
        subprocess.check_call(["make", "-C", CSRC, "-s", "clean"])
    subprocess.check_call(args)
    return LIB_PATH


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: the HIP extension must be built (python -c 'import __graft_entry__ as g; g.build()'); "
                "there is no CPU fallback"
            )
        L = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(L, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        if L.spiral_gpu_abi_version() != 1:
            raise RuntimeError("libspiral_gpu.so ABI version mismatch")
        _lib = L
    return _lib


class SpiralGpuError(RuntimeError):
    pass


def check(rc: int) -> None:
    if rc != 0:
        raise SpiralGpuError(lib().spiral_gpu_last_error().decode() or f"error {rc}")

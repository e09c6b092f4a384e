"""ctypes binding of libslp_hip.so (include/slp_hip.h).

The library is the only execution path: if it is missing, cannot be loaded, or
finds no HIP device, every solver entry point raises -- there is no CPU
fallback in this package.
"""
import ctypes
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SLP_LIB_VARIANT=<name>: libslp_hip_<name>.so -- the -DSLP_ABLATION build (`make -C pysparselp_amd/csrc ablation`, timing
# experiments with wrong results) or a kernel-lab build (`make variant NAME=... EXTRA=...`); never set in production
_VARIANT = os.environ.get("SLP_LIB_VARIANT")
LIB_PATH = os.path.join(_HERE, f"libslp_hip_{_VARIANT}.so" if _VARIANT else "libslp_hip.so")

ORDER_AUTO, ORDER_SEQUENTIAL, ORDER_TREE = 0, 1, 2

_lib = None
_device = None

c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_dbl = ctypes.c_double
c_vp = ctypes.c_void_p

# name -> (restype, argtypes); mirrors include/slp_hip.h one to one
_SIGNATURES = {
    "slp_version": (c_int, []),
    "slp_build_flags": (c_int, []),
    "slp_device_count": (c_int, []),
    "slp_init": (c_int, [c_int]),
    "slp_synchronize": (c_int, []),
    "slp_trim": (c_int, []),
    "slp_cached_bytes": (c_i64, []),
    "slp_alloc_stats": (c_int, [c_vp, c_int]),
    "slp_last_error": (ctypes.c_char_p, []),
    "slp_timer_start": (c_int, []),
    "slp_timer_stop": (c_int, [c_vp]),
    "slp_matrix_create": (c_vp, [c_i64, c_i64, c_vp, c_vp, c_vp]),
    "slp_matrix_destroy": (None, [c_vp]),
    "slp_matrix_nnz": (c_i64, [c_vp]),
    "slp_matrix_spmv": (c_int, [c_vp, c_vp, c_vp, c_int]),
    "slp_matrix_spmv_t": (c_int, [c_vp, c_vp, c_vp, c_int]),
    "slp_matrix_spmv_abs_pow": (c_int, [c_vp, c_int, c_dbl, c_vp, c_vp]),
    "slp_matrix_download": (c_int, [c_vp, c_int, c_vp, c_vp, c_vp]),
    "slp_matrix_download_rows": (c_int, [c_vp, c_int, c_i64, c_i64, c_vp, c_vp, c_vp]),
    "slp_matrix_bench_spmv": (c_int, [c_vp, c_int, c_int, c_int, c_vp]),
    "slp_product_timing": (c_int, [c_int]),
    "slp_product_timing_read": (c_int, [c_vp]),
    "slp_matrix_gather_rows": (c_vp, [c_vp, c_i64, c_vp, c_vp]),
    "slp_matrix_spmv_kernel": (c_int, [c_vp, c_int]),
    "slp_matrix_set_format": (c_int, [c_vp, c_int]),
    "slp_matrix_release_csr": (c_int, [c_vp]),
    "slp_device_memory": (c_int, [c_vp, c_vp]),
    "slp_matrix_normal": (c_vp, [c_vp, c_dbl, c_dbl]),
    "slp_matrix_remove_columns": (c_vp, [c_vp, c_vp, c_vp, c_vp]),
    "slp_matrix_format_bytes": (c_i64, [c_vp, c_int]),
    "slp_matrix_chunked_create": (c_vp, [c_i64]),
    "slp_matrix_chunked_append": (c_int, [c_vp, c_vp]),
    "slp_matrix_chunks": (c_i64, [c_vp]),
    "slp_matrix_product_launches": (c_i64, [c_vp, c_int]),
    "slp_matrix_strip_width": (c_i64, [c_vp, c_int]),
    "slp_matrix_chunked_expect": (c_int, [c_vp, c_i64]),
    "slp_matrix_chunked_expect_rows": (c_int, [c_vp, c_i64, c_i64]),
    "slp_cp_create": (c_vp, [c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int]),
    "slp_cp_create_on": (c_vp, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int]),
    "slp_cp_destroy": (None, [c_vp]),
    "slp_cp_iterate": (c_int, [c_vp, c_i64]),
    "slp_cp_primal_step": (c_int, [c_vp]),
    "slp_cp_split_form": (c_int, [c_vp]),
    "slp_cp_dual_step": (c_int, [c_vp]),
    "slp_cp_report": (c_int, [c_vp, c_vp]),
    "slp_cp_get_x": (c_int, [c_vp, c_vp]),
    "slp_cp_get_y": (c_int, [c_vp, c_vp]),
    "slp_cp_get_preconditioners": (c_int, [c_vp, c_vp, c_vp]),
    "slp_cp_bench": (c_int, [c_vp, c_i64, c_vp]),
    "slp_gs_create": (c_vp, [c_i64, c_vp, c_vp, c_vp]),
    "slp_gs_destroy": (None, [c_vp]),
    "slp_gs_num_levels": (c_i64, [c_vp]),
    "slp_gs_sweep_kind": (c_int, [c_vp]),
    "slp_gs_num_bands": (c_int, [c_vp]),
    "slp_gs_solve": (c_int, [c_vp, c_vp, c_vp, c_vp, c_vp, c_int, c_dbl]),
    "slp_admm_create": (c_vp, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int]),
    "slp_admm_create_lp": (c_vp, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl,
                                  c_int, c_int]),
    "slp_matrix_precondition_rows": (c_vp, [c_vp, c_vp, c_vp]),
    "slp_matrix_standard_form": (c_vp, [c_vp, c_vp]),
    "slp_admm_destroy": (None, [c_vp]),
    "slp_admm_set_xstep": (c_int, [c_vp, c_int]),
    "slp_admm_iterate": (c_int, [c_vp, c_i64]),
    "slp_admm_sweep_step": (c_int, [c_vp]),
    "slp_admm_multiplier_step": (c_int, [c_vp]),
    "slp_admm_report": (c_int, [c_vp, c_vp]),
    "slp_admm_get_x": (c_int, [c_vp, c_vp, c_i64]),
    "slp_admm_get_lambda": (c_int, [c_vp, c_vp]),
    "slp_admm_num_levels": (c_i64, [c_vp]),
    "slp_admm_num_bands": (c_int, [c_vp]),
    "slp_admm_bench": (c_int, [c_vp, c_i64, c_vp]),
    "slp_admm_cg_create": (c_vp, [c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int]),
    "slp_admm_cg_create_on": (c_vp, [c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int]),
    "slp_admm_cg_create_on_mixed": (c_vp, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int]),
    "slp_blocks_create": (c_vp, [c_i64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl]),
    "slp_blocks_create_on": (c_vp, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl]),
    "slp_blocks_group_link": (c_int, [c_vp, c_int]),
    "slp_blocks_group_iterate": (c_int, [c_vp, c_int, c_i64]),
    "slp_blocks_destroy": (None, [c_vp]),
    "slp_blocks_projection_residual": (c_int, [c_vp, c_vp]),
    "slp_blocks_set_cg": (c_int, [c_vp, c_dbl, c_int]),
    "slp_blocks_set_precond": (c_int, [c_vp, c_int]),
    "slp_blocks_iterate": (c_int, [c_vp, c_i64]),
    "slp_blocks_report": (c_int, [c_vp, c_vp]),
    "slp_blocks_cg_steps": (c_i64, [c_vp]),
    "slp_blocks_get_xp": (c_int, [c_vp, c_vp, c_i64]),
    "slp_admm_cg_create_on_two_sided": (c_vp, [c_vp, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp, c_dbl, c_dbl, c_int]),
    "slp_admm_cg_destroy": (None, [c_vp]),
    "slp_admm_cg_set_reuse": (c_int, [c_vp, c_int]),
    "slp_admm_cg_iterate": (c_int, [c_vp, c_i64]),
    "slp_admm_cg_xstep": (c_int, [c_vp]),
    "slp_admm_cg_multiplier_step": (c_int, [c_vp]),
    "slp_admm_cg_report": (c_int, [c_vp, c_vp]),
    "slp_admm_cg_get_x": (c_int, [c_vp, c_vp, c_i64]),
    "slp_matrix_random": (c_vp, [c_i64, c_i64, c_dbl, ctypes.c_uint64, c_i64]),
    "slp_random_lp_vectors": (c_int, [c_vp, c_dbl, ctypes.c_uint64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "slp_random_lp_vectors_eq": (c_int, [c_vp, c_dbl, ctypes.c_uint64, c_i64, c_i64, c_vp, c_vp, c_vp, c_vp, c_vp]),
    "slp_comm_unique_id": (c_int, [c_vp]),
    "slp_comm_init": (c_int, [c_int, c_int, c_vp]),
    "slp_comm_init_host": (c_int, [c_int, c_int, c_vp, c_vp]),
    "slp_comm_finalize": (c_int, []),
    "slp_comm_allreduce_host": (c_int, [c_vp, c_i64, c_int]),
    "slp_comm_barrier": (c_int, []),
    "slp_comm_bench_allreduce": (c_int, [c_i64, c_int, c_vp]),
    "slp_comm_info": (c_int, [c_vp, c_vp]),
    "slp_comm_collectives": (ctypes.c_longlong, []),
    "slp_comm_timing": (c_int, [c_int]),
    "slp_comm_timing_read": (c_int, [c_vp]),
}

EXPORTED_SYMBOLS = tuple(_SIGNATURES)

# int fn(double *buf, int64_t count, int op, void *user): the host all-reduce of slp_comm_init_host
HOST_ALLREDUCE_FN = ctypes.CFUNCTYPE(c_int, ctypes.POINTER(c_dbl), c_i64, c_int, c_vp)


class SlpError(RuntimeError):
    """Raised for every failure reported by libslp_hip.so (and when it is absent)."""


def load():
    """dlopen the library and declare the prototypes (no GPU needed for this)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise SlpError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C pysparselp_amd/csrc`.  pysparselp_amd has no CPU fallback."
        )
    try:
        lib = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise SlpError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype = res
        fn.argtypes = args
    flags = int(lib.slp_build_flags())
    if flags and not _VARIANT:
        raise SlpError(f"{LIB_PATH} is a kernel-lab build (slp_build_flags() = {flags}: bit 0 = -DSLP_ABLATION kernels with parts removed, "
                       "WRONG results; bit 1 = `make variant`): it is only loaded when SLP_LIB_VARIANT names it.  Rebuild with "
                       "`make -C pysparselp_amd/csrc clean all`.")
    _lib = lib
    return lib


def last_error():
    msg = load().slp_last_error()
    return msg.decode("utf-8", "replace") if msg else "unknown error"


def check(rc):
    if rc != 0:
        raise SlpError(last_error())


def check_handle(h):
    if not h:
        raise SlpError(last_error())
    return h


def lib(device=None):
    """The loaded library, bound to a GPU (slp_init).  Raises SlpError when no
    HIP device is usable."""
    global _device
    l = load()
    if _device is None:
        dev = device
        if dev is None:
            dev = int(os.environ.get("SLP_DEVICE", os.environ.get("LOCAL_RANK", "0")))
            n = l.slp_device_count()
            if n > 0 and dev >= n:  # never wrap: that would silently put two ranks on one GPU
                raise SlpError(f"device index {dev} (SLP_DEVICE / LOCAL_RANK) but only {n} HIP device(s) are visible")
        check(l.slp_init(int(dev)))
        _device = int(dev)
    elif device is not None and int(device) != _device:
        raise SlpError(f"process already bound to device {_device}")
    return l


def ptr(a):
    """Device-call pointer of a C-contiguous numpy array (None -> NULL)."""
    if a is None:
        return None
    assert a.flags.c_contiguous
    return a.ctypes.data_as(c_vp)


def f64(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def csr_arrays(a):
    """(indptr int64, indices int32, data float64) of a scipy CSR matrix, entry order untouched."""
    return (np.ascontiguousarray(a.indptr, dtype=np.int64), np.ascontiguousarray(a.indices, dtype=np.int32),
            np.ascontiguousarray(a.data, dtype=np.float64))

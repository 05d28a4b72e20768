"""ctypes binding of libcsr5hip.so -- the C ABI declared in include/csr5hip.h.

There is no CPU fallback: if the HIP library is missing or cannot be loaded, importing the product
path raises.  Build it with ``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C benchmark_spmv_using_csr5_amd/csrc``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libcsr5hip.so")

SUCCESS = 0
UNKOWN_FORMAT = -1
UNSUPPORTED_CSR5_OMEGA = -2
CSR_TO_CSR5_FAILED = -3
UNSUPPORTED_CSR_SPMV = -4
UNSUPPORTED_VALUE_TYPE = -5
HIP_ERROR = -100
INVALID_ARGUMENT = -101

FORMAT_CSR = 0
FORMAT_CSR5 = 1
OMEGA = 64
AUTO_TUNED_SIGMA = -1
F64 = 0
F32 = 1
OPT_SPMV_MODE = 1
OPT_XCD_REMAP = 2
OPT_X_WINDOW = 3
OPT_LDS_Y = 4
OPT_STREAM_NT = 5
OPT_COLUMN_SLABS = 6
OPT_SLAB_SHIFT = 7
OPT_ZERO_EMPTY_ROWS = 8
OPT_SLAB_HOT = 9
OPT_SLAB_MEMORY_MIB = 10
OPT_X_SNAPSHOT = 11
OPT_NARROW_VALUES = 12
OPT_NARROW_COLUMNS = 15
OPT_DEFER_CARRIES = 16
OPT_FLAGGED_COLUMNS = 18
MULTI_OPT_ROW_WEIGHT = 100  # csr5hip_multi_set_option only (before input_csr)
MULTI_OPT_OWN_REPLICAS = 101  # csr5hip_multi_set_option only (before set_x): devices[0]'s shards read a broadcast replica too


class Csr5Info(C.Structure):
    _fields_ = [
        ("format", C.c_int), ("m", C.c_int), ("n", C.c_int), ("nnz", C.c_int),
        ("value_type", C.c_int), ("omega", C.c_int), ("sigma", C.c_int),
        ("bit_y_offset", C.c_int), ("bit_scansum_offset", C.c_int), ("num_packet", C.c_int),
        ("p", C.c_int), ("tail_partition_start", C.c_int), ("num_offsets", C.c_int),
        ("d_tile_ptr", C.c_void_p), ("d_tile_desc", C.c_void_p),
        ("d_offset_ptr", C.c_void_p), ("d_offset", C.c_void_p),
        ("x_window_tiles", C.c_int), ("x_window_active", C.c_int), ("x_window_cover_pct", C.c_int),
        ("x_window_lines", C.c_int),
        ("t_malloc_ms", C.c_double), ("t_tile_ptr_ms", C.c_double),
        ("t_tile_desc_ms", C.c_double), ("t_transpose_ms", C.c_double),
        ("column_slabs", C.c_int), ("slab_shift", C.c_int), ("slab_segments", C.c_int),
        ("slab_sigma", C.c_int), ("slab_tiles", C.c_int), ("t_slab_ms", C.c_double),
        ("slab_hot", C.c_int), ("slab_hot_cover_pct", C.c_int), ("slab_fallback", C.c_int),
        ("device_bytes", C.c_longlong),
        ("slab_x_permuted", C.c_int), ("slab_cold_entries", C.c_int), ("x_snapshot", C.c_int),
        ("slab_values_narrowed", C.c_int),
        ("carries_deferred", C.c_int),
        ("narrow_columns", C.c_int),
        ("flagged_columns", C.c_int),
    ]


MTX_CANNOT_OPEN = -1
MTX_BAD_BANNER = -2
MTX_COMPLEX = -3
MTX_BAD_SIZE = -4
FIELD_REAL, FIELD_INTEGER, FIELD_PATTERN = 0, 1, 2


class MtxCoo(C.Structure):  # csr5hip_mtx
    _fields_ = [
        ("m", C.c_int32), ("n", C.c_int32), ("nz", C.c_int64), ("field", C.c_int), ("symmetric", C.c_int),
        ("row", C.POINTER(C.c_int32)), ("col", C.POINTER(C.c_int32)), ("val", C.POINTER(C.c_double)),
        ("threads", C.c_int), ("fast_path", C.c_int), ("t_parse_ms", C.c_double), ("file_bytes", C.c_int64),
        ("alloc_flags", C.c_int),
    ]


class DeviceCsrStruct(C.Structure):  # csr5hip_csr
    _fields_ = [
        ("m", C.c_int32), ("n", C.c_int32), ("nnz", C.c_int32),
        ("d_row_ptr", C.c_void_p), ("d_col_idx", C.c_void_p), ("d_val", C.c_void_p),
        ("value_type", C.c_int),
        ("t_parse_ms", C.c_double), ("t_h2d_ms", C.c_double), ("t_build_ms", C.c_double),
    ]


class Shard(C.Structure):  # csr5hip_shard
    _fields_ = [("device", C.c_int), ("row_lo", C.c_int), ("row_hi", C.c_int), ("nnz", C.c_int),
                ("d_y", C.c_void_p), ("handle", C.c_void_p), ("x_broadcast", C.c_int)]


# every symbol include/csr5hip.h declares: (name, restype, argtypes)
_H = C.c_void_p
SYMBOLS = [
    ("csr5hip_create", C.c_int, [C.POINTER(_H), C.c_int, C.c_int, C.c_int]),
    ("csr5hip_free", C.c_int, [_H]),
    ("csr5hip_set_stream", C.c_int, [_H, C.c_void_p]),
    ("csr5hip_warmup", C.c_int, [_H]),
    ("csr5hip_input_csr", C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("csr5hip_set_x", C.c_int, [_H, C.c_void_p]),
    ("csr5hip_set_sigma", C.c_int, [_H, C.c_int]),
    ("csr5hip_as_csr5", C.c_int, [_H]),
    ("csr5hip_as_csr", C.c_int, [_H]),
    ("csr5hip_spmv", C.c_int, [_H, C.c_double, C.c_void_p]),
    ("csr5hip_spmv_repeat", C.c_int, [_H, C.c_double, C.c_void_p, C.c_int]),
    ("csr5hip_spmv_rotate", C.c_int, [C.POINTER(_H), C.POINTER(C.c_void_p), C.c_int, C.c_double, C.c_int]),
    ("csr5hip_snapshot_x", C.c_int, [_H]),
    ("csr5hip_destroy", C.c_int, [_H]),
    ("csr5hip_autotune_sigma", C.c_int, [_H, C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]),
    ("csr5hip_set_option", C.c_int, [_H, C.c_int, C.c_int]),
    ("csr5hip_get_info", C.c_int, [_H, C.POINTER(Csr5Info)]),
    ("csr5hip_auto_sigma", C.c_int, [C.c_int, C.c_int, C.c_int]),
    ("csr5hip_last_error", C.c_char_p, []),
    ("csr5hip_version", C.c_char_p, []),
    ("csr5hip_device_count", C.c_int, [C.POINTER(C.c_int)]),
    ("csr5hip_set_device", C.c_int, [C.c_int]),
    ("csr5hip_device_name", C.c_int, [C.c_int, C.c_char_p, C.c_size_t, C.POINTER(C.c_double)]),
    ("csr5hip_malloc", C.c_int, [C.POINTER(C.c_void_p), C.c_size_t]),
    ("csr5hip_device_free", C.c_int, [C.c_void_p]),
    ("csr5hip_memcpy_h2d", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("csr5hip_memcpy_d2h", C.c_int, [C.c_void_p, C.c_void_p, C.c_size_t]),
    ("csr5hip_memset", C.c_int, [C.c_void_p, C.c_int, C.c_size_t]),
    ("csr5hip_synchronize", C.c_int, []),
    ("csr5hip_timer_start", C.c_int, [_H]),
    ("csr5hip_timer_stop", C.c_int, [_H, C.POINTER(C.c_double)]),
    ("csr5hip_mtx_read", C.c_int, [C.c_char_p, C.c_int, C.POINTER(MtxCoo)]),
    ("csr5hip_mtx_release", C.c_int, [C.POINTER(MtxCoo)]),
    ("csr5hip_coo_to_csr", C.c_int, [C.c_int32, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p,
                                     C.c_int, C.c_int, C.POINTER(DeviceCsrStruct)]),
    ("csr5hip_csr_release", C.c_int, [C.POINTER(DeviceCsrStruct)]),
    ("csr5hip_mtx_load", C.c_int, [C.c_char_p, C.c_int, C.c_int, C.POINTER(DeviceCsrStruct)]),
    ("csr5hip_multi_create", C.c_int, [C.POINTER(_H), C.POINTER(C.c_int), C.c_int, C.c_int, C.c_int, C.c_int]),
    ("csr5hip_multi_free", C.c_int, [_H]),
    ("csr5hip_multi_input_csr", C.c_int, [_H, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]),
    ("csr5hip_multi_set_sigma", C.c_int, [_H, C.c_int]),
    ("csr5hip_multi_set_option", C.c_int, [_H, C.c_int, C.c_int]),
    ("csr5hip_multi_as_csr5", C.c_int, [_H]),
    ("csr5hip_multi_set_x", C.c_int, [_H, C.c_void_p]),
    ("csr5hip_multi_spmv", C.c_int, [_H, C.c_double]),
    ("csr5hip_multi_spmv_repeat", C.c_int, [_H, C.c_double, C.c_int]),
    ("csr5hip_multi_synchronize", C.c_int, [_H]),
    ("csr5hip_multi_timer_start", C.c_int, [_H]),
    ("csr5hip_multi_timer_stop", C.c_int, [_H, C.POINTER(C.c_double)]),
    ("csr5hip_multi_shard", C.c_int, [_H, C.c_int, C.POINTER(Shard)]),
    ("csr5hip_multi_gather_y", C.c_int, [_H, C.c_void_p]),
    ("csr5hip_multi_fill_y", C.c_int, [_H, C.c_int]),
    ("csr5hip_multi_destroy", C.c_int, [_H]),
    ("csr5hip_save", C.c_int, [_H, C.c_char_p]),
    ("csr5hip_load", C.c_int, [C.c_char_p, C.POINTER(_H), C.POINTER(DeviceCsrStruct)]),
]

_lib = None


def load():
    """Load libcsr5hip.so (once).  Raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("CSR5HIP_LIB", LIB_PATH)  # experiment builds only (timing / ablation)
    if not os.path.exists(path):
        raise ImportError(
            f"{path} is missing: the CSR5 HIP extension is not built "
            "(run `python -c 'import __graft_entry__ as g; g.build()'`). There is no CPU fallback.")
    # One HIP runtime per process: torch's libtorch_hip.so asks for "libamdhip64.so" and finds its bundled
    # copy through RPATH even when /opt/rocm's libamdhip64.so.7 is already mapped, and the second runtime
    # then sees no GPU.  Loading torch first makes our NEEDED libamdhip64.so.7 resolve to the copy torch
    # brought in (same SONAME).  Programs that never use torch (the ./spmv CLI, C hosts) are unaffected.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(path)
    for name, restype, argtypes in SYMBOLS:
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def last_error() -> str:
    return load().csr5hip_last_error().decode()

"""Matrix Market ingest and COO -> CSR through libcsr5hip.so.

Python mirror of what the reference CLI does before it reaches the handle (CSR5_avx2/main.cpp:126-281):
``read_mtx_coo`` is the multi-threaded host parser (needs no GPU), ``coo_to_csr`` / ``load_mtx`` build the
CSR on the device.  The CSR is identical, entry for entry, to the one the reference's serial loops build.
There is no CPU fallback for the device part.
"""
from __future__ import annotations

import ctypes as C
import os
from dataclasses import dataclass, field

import numpy as np

from . import _capi
from .matrices import CsrMatrix, MtxError


@dataclass
class MtxCoo:
    """COO triplets of a .mtx file, 0-based, file order (host copies)."""

    m: int
    n: int
    nz: int
    field: int            # _capi.FIELD_*
    symmetric: bool
    row: np.ndarray = field(repr=False)
    col: np.ndarray = field(repr=False)
    val: np.ndarray = field(repr=False)
    threads: int = 1
    fast_path: bool = True
    parse_ms: float = 0.0
    file_bytes: int = 0


def _raise(rc: int, what: str):
    msg = _capi.last_error()
    if rc in (_capi.MTX_CANNOT_OPEN, _capi.MTX_BAD_BANNER, _capi.MTX_COMPLEX, _capi.MTX_BAD_SIZE):
        raise MtxError(rc, msg or what)
    if rc == _capi.INVALID_ARGUMENT:
        raise ValueError(f"{what}: {msg}")
    raise RuntimeError(f"{what} failed with {rc}: {msg}")


def read_mtx_coo(path: str, threads: int = 0) -> MtxCoo:
    """Parse `path` with the library's parallel parser.  Raises MtxError(code) with the CLI's exit codes."""
    lib = _capi.load()
    raw = _capi.MtxCoo()
    rc = lib.csr5hip_mtx_read(os.fsencode(path), int(threads), C.byref(raw))
    if rc:
        _raise(rc, "csr5hip_mtx_read")
    try:
        nz = int(raw.nz)
        if nz:
            row = np.ctypeslib.as_array(raw.row, shape=(nz,)).copy()
            col = np.ctypeslib.as_array(raw.col, shape=(nz,)).copy()
            val = np.ctypeslib.as_array(raw.val, shape=(nz,)).copy()
        else:
            row = np.zeros(0, np.int32)
            col = np.zeros(0, np.int32)
            val = np.zeros(0, np.float64)
        return MtxCoo(int(raw.m), int(raw.n), nz, int(raw.field), bool(raw.symmetric), row, col, val,
                      int(raw.threads), bool(raw.fast_path), float(raw.t_parse_ms), int(raw.file_bytes))
    finally:
        lib.csr5hip_mtx_release(C.byref(raw))


class DeviceCsr:
    """CSR arrays in HBM, allocated by the library (csr5hip_csr).  Pointers can go straight into
    ``anonymouslibHandle.inputCSR``.  ``release()`` (or garbage collection) frees them."""

    def __init__(self, raw: _capi.DeviceCsrStruct):
        self._raw = raw
        self.m, self.n, self.nnz = int(raw.m), int(raw.n), int(raw.nnz)
        self.value_type = int(raw.value_type)
        self.parse_ms, self.h2d_ms, self.build_ms = float(raw.t_parse_ms), float(raw.t_h2d_ms), float(raw.t_build_ms)

    @property
    def row_ptr(self) -> int:
        return int(self._raw.d_row_ptr or 0)

    @property
    def col_idx(self) -> int:
        return int(self._raw.d_col_idx or 0)

    @property
    def val(self) -> int:
        return int(self._raw.d_val or 0)

    @property
    def dtype(self):
        return np.float64 if self.value_type == _capi.F64 else np.float32

    def _d2h(self, dptr: int, count: int, dtype) -> np.ndarray:
        out = np.zeros(count, dtype=dtype)
        if count:
            rc = _capi.load().csr5hip_memcpy_d2h(out.ctypes.data, dptr, out.nbytes)
            if rc:
                raise RuntimeError(f"csr5hip_memcpy_d2h failed: {_capi.last_error()}")
        return out

    def to_host(self, name: str = "") -> CsrMatrix:
        row_ptr = self._d2h(self.row_ptr, self.m + 1, np.int32)
        col = self._d2h(self.col_idx, self.nnz, np.int32)
        val = self._d2h(self.val, self.nnz, self.dtype) if self.val else np.zeros(self.nnz, self.dtype)
        return CsrMatrix(self.m, self.n, row_ptr, col, val, name=name)

    def release(self) -> None:
        if self._raw is not None:
            _capi.load().csr5hip_csr_release(C.byref(self._raw))
            self._raw = None

    def __del__(self):
        try:
            self.release()
        except Exception:
            pass


def coo_to_csr(m: int, n: int, row, col, val, symmetric: bool, dtype=np.float64) -> DeviceCsr:
    """row / col: torch CUDA int32 tensors with the COO triplets in file order; val: float64 tensor or None
    (structure only).  Returns the CSR in HBM (library-allocated arrays)."""
    lib = _capi.load()
    nz = int(row.numel())
    vt = _capi.F64 if np.dtype(dtype) == np.float64 else _capi.F32
    raw = _capi.DeviceCsrStruct()
    rc = lib.csr5hip_coo_to_csr(int(m), int(n), nz, row.data_ptr() if nz else None, col.data_ptr() if nz else None,
                                (val.data_ptr() if nz else None) if val is not None else None,
                                int(bool(symmetric)), vt, C.byref(raw))
    if rc:
        _raise(rc, "csr5hip_coo_to_csr")
    return DeviceCsr(raw)


def load_mtx(path: str, dtype=np.float64, threads: int = 0) -> DeviceCsr:
    """Parse + H2D + device COO->CSR in one call (what ``./spmv foo.mtx`` does first)."""
    lib = _capi.load()
    vt = _capi.F64 if np.dtype(dtype) == np.float64 else _capi.F32
    raw = _capi.DeviceCsrStruct()
    rc = lib.csr5hip_mtx_load(os.fsencode(path), int(threads), vt, C.byref(raw))
    if rc:
        _raise(rc, "csr5hip_mtx_load")
    return DeviceCsr(raw)

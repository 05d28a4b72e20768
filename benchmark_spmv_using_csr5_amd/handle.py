"""Python mirror of the reference's ``anonymouslibHandle<int, unsigned int, VALUE_TYPE>``.

Same member names, argument meaning, state machine and integer return codes as
CSR5_cuda/anonymouslib_cuda.h:11-53; every call goes straight through the C ABI of libcsr5hip.so
(include/csr5hip.h).  Arguments are *device* arrays owned by the caller -- here ``torch`` CUDA tensors
(torch is plumbing for device memory and streams only) or raw integer device pointers.  ``asCSR5``
permutes the caller's ``col_idx`` / ``val`` tensors in place, exactly as the reference does.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _capi

ANONYMOUSLIB_SUCCESS = _capi.SUCCESS
ANONYMOUSLIB_UNSUPPORTED_CSR5_OMEGA = _capi.UNSUPPORTED_CSR5_OMEGA
ANONYMOUSLIB_CSR_TO_CSR5_FAILED = _capi.CSR_TO_CSR5_FAILED
ANONYMOUSLIB_UNSUPPORTED_CSR_SPMV = _capi.UNSUPPORTED_CSR_SPMV
ANONYMOUSLIB_UNSUPPORTED_VALUE_TYPE = _capi.UNSUPPORTED_VALUE_TYPE
ANONYMOUSLIB_FORMAT_CSR = _capi.FORMAT_CSR
ANONYMOUSLIB_FORMAT_CSR5 = _capi.FORMAT_CSR5
ANONYMOUSLIB_CSR5_OMEGA = _capi.OMEGA
ANONYMOUSLIB_AUTO_TUNED_SIGMA = _capi.AUTO_TUNED_SIGMA

SPMV_TWO_PASS = 0
SPMV_FUSED = 1


def _ptr(t) -> int:
    if t is None:
        return 0
    if isinstance(t, int):
        return t
    return int(t.data_ptr())


def _value_type(dtype) -> int:
    name = str(dtype)
    if name.endswith("float64"):
        return _capi.F64
    if name.endswith("float32"):
        return _capi.F32
    raise TypeError(f"unsupported VALUE_TYPE {dtype}: only float64 and float32 "
                    "(README.md:71 of the reference)")


class anonymouslibHandle:
    """``A = anonymouslibHandle(m, n, dtype)``; then ``inputCSR / setX / setSigma / asCSR5 / spmv /
    destroy`` as in CSR5_cuda/main.cu:59-104."""

    def __init__(self, m: int, n: int, dtype="float64", stream=None):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        self._vt = _value_type(dtype)
        err = self._lib.csr5hip_create(C.byref(self._h), int(m), int(n), self._vt)
        if err:
            raise RuntimeError(f"csr5hip_create -> {err}")
        self._keep = {}  # borrowed tensors, kept alive while the handle points at them
        if stream is not None:
            self.setStream(stream)

    # -- reference API ------------------------------------------------------------------------
    def warmup(self) -> int:
        return self._lib.csr5hip_warmup(self._h)

    def inputCSR(self, nnz: int, csr_row_pointer, csr_column_index, csr_value) -> int:
        self._keep.update(row_ptr=csr_row_pointer, col=csr_column_index, val=csr_value)
        return self._lib.csr5hip_input_csr(self._h, int(nnz), _ptr(csr_row_pointer),
                                           _ptr(csr_column_index), _ptr(csr_value))

    def setX(self, x) -> int:
        self._keep.update(x=x)
        return self._lib.csr5hip_set_x(self._h, _ptr(x))

    def setSigma(self, sigma: int) -> int:
        return self._lib.csr5hip_set_sigma(self._h, int(sigma))

    def asCSR5(self) -> int:
        return self._lib.csr5hip_as_csr5(self._h)

    def asCSR(self) -> int:
        return self._lib.csr5hip_as_csr(self._h)

    def spmv(self, alpha, y) -> int:
        return self._lib.csr5hip_spmv(self._h, float(alpha), _ptr(y))

    def destroy(self) -> int:
        return self._lib.csr5hip_destroy(self._h)

    # -- additions (documented in include/csr5hip.h) ---------------------------------------------
    def spmv_repeat(self, alpha, y, count: int) -> int:
        return self._lib.csr5hip_spmv_repeat(self._h, float(alpha), _ptr(y), int(count))

    def snapshotX(self) -> int:
        """X_SNAPSHOT mode: take the permuted copy of x now, on the handle's stream (csr5hip.h csr5hip_snapshot_x)"""
        return self._lib.csr5hip_snapshot_x(self._h)

    @staticmethod
    def spmv_rotate(handles, ys, count: int) -> int:
        """`count` SpMVs from one hipGraph, the i-th on handles[i % k] into ys[i % k] (cold-cache protocol, csr5hip.h)."""
        k = len(handles)
        hs = (C.c_void_p * k)(*[h._h.value for h in handles])
        yp = (C.c_void_p * k)(*[_ptr(y) for y in ys])
        return handles[0]._lib.csr5hip_spmv_rotate(hs, yp, k, 1.0, int(count))

    def autotuneSigma(self, y):
        """Measured sigma selection: returns (err, sigma, us_per_spmv); leaves the matrix in CSR5."""
        sigma, us = C.c_int(0), C.c_double(0.0)
        err = self._lib.csr5hip_autotune_sigma(self._h, _ptr(y), C.byref(sigma), C.byref(us))
        return err, sigma.value, us.value

    def setStream(self, stream) -> int:
        raw = getattr(stream, "cuda_stream", stream)
        return self._lib.csr5hip_set_stream(self._h, C.c_void_p(int(raw) if raw else None))

    def setOption(self, option: int, value: int) -> int:
        return self._lib.csr5hip_set_option(self._h, int(option), int(value))

    def setSpmvMode(self, mode: int) -> int:
        return self.setOption(_capi.OPT_SPMV_MODE, mode)

    def setXWindow(self, value: int) -> int:
        """0 = off, 1 = auto (default), 2 = force the LDS x-window variant of the fused kernel."""
        return self.setOption(_capi.OPT_X_WINDOW, value)

    def setLdsY(self, value: int) -> int:
        """0 = off, 1 = auto (default), 2 = force the LDS compaction of a tile's y segments."""
        return self.setOption(_capi.OPT_LDS_Y, value)

    def setStreamNT(self, value: int) -> int:
        """0 off, 1 auto (non-temporal column/value loads when the streams exceed the Infinity Cache), 2 force"""
        return self.setOption(_capi.OPT_STREAM_NT, int(value))

    def setColumnSlabs(self, value: int) -> int:
        """0 off, 1 auto (default), 2..64 (power of two) = that many column slabs (kernel-side structure, csr5hip.h)"""
        return self.setOption(_capi.OPT_COLUMN_SLABS, int(value))

    def setSlabShift(self, value: int) -> int:
        return self.setOption(_capi.OPT_SLAB_SHIFT, int(value))

    def setSlabHot(self, value: int) -> int:
        """column slabs: 0 = no LDS hot table, 1 = auto (default), 2 = force (csr5hip.h CSR5HIP_OPT_SLAB_HOT)"""
        return self.setOption(_capi.OPT_SLAB_HOT, int(value))

    def setSlabMemoryMiB(self, value: int) -> int:
        """upper bound (MiB) on the device memory of the column-slab structure, 0 = none: a structure that would not fit
        is not built and spmv() runs the plain kernel (info().slab_fallback == 1)"""
        return self.setOption(_capi.OPT_SLAB_MEMORY_MIB, int(value))

    def setXSnapshot(self, value: int) -> int:
        """hot-table slab kernel: 0 (default) = its permuted copy of x is refreshed by every spmv() (x read live, as the
        reference does); 1 = once per setX() -- the caller promises not to change x's contents in between (what the
        reference CLI does: CSR5_cuda/main.cu:63); csr5hip.h CSR5HIP_OPT_X_SNAPSHOT"""
        return self.setOption(_capi.OPT_X_SNAPSHOT, int(value))

    def setNarrowValues(self, value: int) -> int:
        """fp64 + hot table: 1 = stream the values as fp32 when every one of them is exactly representable (same results bit
        for bit, 4 bytes less per non-zero); 0 (default) = off; csr5hip.h CSR5HIP_OPT_NARROW_VALUES"""
        return self.setOption(_capi.OPT_NARROW_VALUES, int(value))

    def setDeferCarries(self, value: int) -> int:
        """cut rows finished by a second small launch instead of inside the SpMV launch (no short-spill re-reads, no arrival
        atomics): 0 = off, 1 = auto (default), 2 = force; set before asCSR5 (csr5hip.h CSR5HIP_OPT_DEFER_CARRIES)"""
        return self.setOption(_capi.OPT_DEFER_CARRIES, int(value))

    def setFlaggedColumns(self, value: int) -> int:
        """plain kernel at sigma 4..8: stream column words that carry the row-start flag in bit 31 (no descriptor load): 1 = auto
        (default: matrices whose streams exceed the Infinity Cache), 0 = off, 2 = force (csr5hip.h CSR5HIP_OPT_FLAGGED_COLUMNS)"""
        return self.setOption(_capi.OPT_FLAGGED_COLUMNS, int(value))

    def setNarrowColumns(self, value: int) -> int:
        """x-window kernel: 1 = auto (default) stream 16-bit column codes (15 bits of column + the row-start flag) when every tile spans < 32 768 columns, 0 = off
        (csr5hip.h CSR5HIP_OPT_NARROW_COLUMNS)"""
        return self.setOption(_capi.OPT_NARROW_COLUMNS, int(value))

    def setZeroEmptyRows(self, value: int) -> int:
        """1 = spmv() also stores 0 into rows without non-zeros (solver coupling); 0 = reference behaviour"""
        return self.setOption(_capi.OPT_ZERO_EMPTY_ROWS, int(value))

    def info(self) -> _capi.Csr5Info:
        info = _capi.Csr5Info()
        err = self._lib.csr5hip_get_info(self._h, C.byref(info))
        if err:
            raise RuntimeError(f"csr5hip_get_info -> {err}")
        return info

    def timer_start(self) -> int:
        return self._lib.csr5hip_timer_start(self._h)

    def timer_stop(self) -> float:
        ms = C.c_double(0.0)
        err = self._lib.csr5hip_timer_stop(self._h, C.byref(ms))
        if err:
            raise RuntimeError(f"csr5hip_timer_stop -> {err}: {_capi.last_error()}")
        return ms.value

    def _d2h(self, dptr, count: int, dtype) -> np.ndarray:
        out = np.empty(count, dtype=dtype)
        if count:
            err = self._lib.csr5hip_memcpy_d2h(out.ctypes.data, dptr, out.nbytes)
            if err:
                raise RuntimeError(f"csr5hip_memcpy_d2h -> {err}: {_capi.last_error()}")
        return out

    def csr5_arrays(self) -> dict:
        """Host copies of the CSR5 auxiliary arrays (for parity tests)."""
        i = self.info()
        if i.format != _capi.FORMAT_CSR5:
            raise RuntimeError("matrix is not in CSR5 format")
        return dict(
            sigma=i.sigma, bit_y=i.bit_y_offset, bit_ss=i.bit_scansum_offset,
            num_packet=i.num_packet, p=i.p, tail_start=i.tail_partition_start,
            num_offsets=i.num_offsets,
            tile_ptr=self._d2h(i.d_tile_ptr, i.p + 1, np.uint32),
            tile_desc=self._d2h(i.d_tile_desc, i.p * i.omega * i.num_packet, np.uint32),
            offset_ptr=self._d2h(i.d_offset_ptr, i.p + 1, np.int32),
            offset=self._d2h(i.d_offset, i.num_offsets, np.int32),
        )

    # -- checkpoint -----------------------------------------------
    def save(self, path: str) -> int:
        """Write the CSR5 state (row_ptr, tile-ordered col/val, the four format arrays) to `path`."""
        import os
        return self._lib.csr5hip_save(self._h, os.fsencode(path))

    @classmethod
    def load(cls, path: str):
        """Restore a checkpoint written by :meth:`save` -> handle already in CSR5 format (no conversion).
        The CSR arrays live in ``handle.arrays`` (an ``ingest.DeviceCsr`` owned by the handle object)."""
        import os
        from .ingest import DeviceCsr
        lib = _capi.load()
        h = C.c_void_p()
        raw = _capi.DeviceCsrStruct()
        err = lib.csr5hip_load(os.fsencode(path), C.byref(h), C.byref(raw))
        if err:
            raise RuntimeError(f"csr5hip_load -> {err}: {_capi.last_error()}")
        self = cls.__new__(cls)
        self._lib, self._h, self._vt, self._keep = lib, h, int(raw.value_type), {}
        self.arrays = DeviceCsr(raw)
        return self

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.csr5hip_free(self._h)
            self._h = C.c_void_p()
        self._keep = {}
        arrays = getattr(self, "arrays", None)
        if arrays is not None:  # CSR arrays of a loaded checkpoint: released after the handle that borrowed them
            arrays.release()
            self.arrays = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class MultiGpuHandle:
    """One matrix on the G GPUs of a node through the C ABI (``csr5hip_multi_*``, include/csr5hip.h): cost-balanced (nnz + 2 * rows) row
    blocks, one ordinary handle + stream per device, x replicated by ONE RCCL broadcast at ``setX``, y sharded.
    ``devices`` may repeat a device id (several shards on one GPU) -- how a 1-GPU box exercises the path."""

    def __init__(self, devices, m: int, n: int, dtype="float64"):
        self._lib = _capi.load()
        self._h = C.c_void_p()
        self.m, self.n, self.G = int(m), int(n), len(devices)
        self._np_dtype = np.float64 if _value_type(dtype) == _capi.F64 else np.float32
        devs = (C.c_int * self.G)(*[int(d) for d in devices])
        self._check(self._lib.csr5hip_multi_create(C.byref(self._h), devs, self.G, self.m, self.n, _value_type(dtype)),
                    "csr5hip_multi_create")
        self._keep = {}

    def _check(self, rc: int, what: str) -> None:
        if rc != 0:
            raise RuntimeError(f"{what} -> {rc}: {_capi.last_error()}")

    def inputCSR(self, nnz, row_ptr, col, val) -> int:
        """device arrays on devices[0]; copied into the shards (the caller's arrays stay untouched)"""
        return self._lib.csr5hip_multi_input_csr(self._h, int(nnz), _ptr(row_ptr), _ptr(col), _ptr(val))

    def setSigma(self, sigma: int) -> int:
        return self._lib.csr5hip_multi_set_sigma(self._h, int(sigma))

    def setOption(self, option: int, value: int) -> int:
        return self._lib.csr5hip_multi_set_option(self._h, int(option), int(value))

    def setRowWeight(self, weight: int) -> int:
        """cost of a row in non-zeros when the row blocks are cut (default 2; 0 = plain nnz balance); before inputCSR"""
        return self.setOption(_capi.MULTI_OPT_ROW_WEIGHT, int(weight))

    def asCSR5(self) -> int:
        return self._lib.csr5hip_multi_as_csr5(self._h)

    def setX(self, x) -> int:
        self._keep["x"] = x
        return self._lib.csr5hip_multi_set_x(self._h, _ptr(x))

    def spmv(self, alpha=1.0) -> int:
        return self._lib.csr5hip_multi_spmv(self._h, float(alpha))

    def spmv_repeat(self, alpha, count: int) -> int:
        return self._lib.csr5hip_multi_spmv_repeat(self._h, float(alpha), int(count))

    def synchronize(self) -> int:
        return self._lib.csr5hip_multi_synchronize(self._h)

    def timer_start(self) -> int:
        return self._lib.csr5hip_multi_timer_start(self._h)

    def timer_stop(self) -> float:
        ms = C.c_double(0.0)
        self._check(self._lib.csr5hip_multi_timer_stop(self._h, C.byref(ms)), "csr5hip_multi_timer_stop")
        return ms.value

    def shard(self, g: int) -> _capi.Shard:
        s = _capi.Shard()
        self._check(self._lib.csr5hip_multi_shard(self._h, int(g), C.byref(s)), "csr5hip_multi_shard")
        return s

    def shard_info(self, g: int) -> _capi.Csr5Info:
        info = _capi.Csr5Info()
        self._check(self._lib.csr5hip_get_info(C.c_void_p(self.shard(g).handle), C.byref(info)), "csr5hip_get_info")
        return info

    def fill_y(self, byte_value: int) -> int:
        return self._lib.csr5hip_multi_fill_y(self._h, int(byte_value))

    def gather_y(self) -> np.ndarray:
        out = np.empty(self.m, dtype=self._np_dtype)
        self._check(self._lib.csr5hip_multi_gather_y(self._h, out.ctypes.data), "csr5hip_multi_gather_y")
        return out

    def destroy(self) -> int:
        return self._lib.csr5hip_multi_destroy(self._h)

    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.csr5hip_multi_free(self._h)
            self._h = C.c_void_p()
        self._keep = {}

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

"""ctypes front-end of the parity oracle (TEST INFRASTRUCTURE ONLY).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
module.  The product package (``benchmark_spmv_using_csr5_amd``) never does.

Two back-ends:

* :class:`Oracle`  -- ``oracle/libcsr5oracle.so``: our plain-C restatement of the reference algorithm
  (``oracle/csr5_oracle.c``), omega/sigma are run-time parameters.
* :class:`Reference` -- ``oracle/_ref/*.so``: the reference's own ``CSR5_avx2`` sources compiled where
  they lie under ``/root/reference`` (``make -C oracle ref``).  Present only when it was built in the
  build container; on the GPU box the prebuilt binaries travel with the repository snapshot.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_I32P = np.ctypeslib.ndpointer(dtype=np.int32, flags="C_CONTIGUOUS")
_U32P = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")
_F64P = np.ctypeslib.ndpointer(dtype=np.float64, flags="C_CONTIGUOUS")
_F32P = np.ctypeslib.ndpointer(dtype=np.float32, flags="C_CONTIGUOUS")

OFFSET_SENTINEL = -1  # value of offset[] slots the reference never writes


@dataclass
class Csr5Format:
    """Host copy of the CSR5 arrays of one matrix (reference names in brackets)."""

    omega: int
    sigma: int
    m: int
    nnz: int
    bit_y: int          # _bit_y_offset
    bit_ss: int         # _bit_scansum_offset
    num_packet: int     # _num_packet
    p: int              # _p
    tail_start: int     # _tail_partition_start
    num_offsets: int    # _num_offsets
    tile_ptr: np.ndarray = field(repr=False)     # _csr5_partition_pointer            uint32[p+1]
    tile_desc: np.ndarray = field(repr=False)    # _csr5_partition_descriptor         uint32[p*omega*num_packet]
    offset_ptr: np.ndarray = field(repr=False)   # _csr5_partition_descriptor_offset_pointer int32[p+1]
    offset: np.ndarray = field(repr=False)       # _csr5_partition_descriptor_offset  int32[num_offsets]
    col: np.ndarray = field(repr=False)          # tile-transposed column_index
    val: np.ndarray = field(repr=False)          # tile-transposed value


class MtxExit(Exception):
    """The exit code the reference CLI would return for an unreadable .mtx (main.cpp:135-157)."""

    def __init__(self, code: int):
        super().__init__(f"mtx ingest exit code {code}")
        self.code = code


@dataclass
class MtxIngest:
    m: int
    n: int
    nnz: int            # after symmetric expansion
    nz_file: int        # entries in the file
    symmetric: bool
    field: int          # 0 real, 1 integer, 2 pattern
    coo_row: np.ndarray = field(repr=False)
    coo_col: np.ndarray = field(repr=False)
    coo_val: np.ndarray = field(repr=False)
    row_ptr: np.ndarray = field(repr=False)
    col: np.ndarray = field(repr=False)
    val: np.ndarray = field(repr=False)


def build_oracle(force: bool = False) -> str:
    """Compile oracle/libcsr5oracle.so (and, where /root/reference exists, oracle/_ref)."""
    so = os.path.join(_HERE, "libcsr5oracle.so")
    srcs = [os.path.join(_HERE, f) for f in ("csr5_oracle.c", "mtx_oracle.c")]
    if force or not os.path.exists(so) or os.path.getmtime(so) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "libcsr5oracle.so"], stdout=subprocess.DEVNULL)
    if os.path.isdir("/root/reference/CSR5_avx2"):
        subprocess.check_call(["make", "-C", _HERE, "ref"], stdout=subprocess.DEVNULL)
    return so


class Oracle:
    def __init__(self) -> None:
        so = os.path.join(_HERE, "libcsr5oracle.so")
        if not os.path.exists(so):
            build_oracle()
        L = C.CDLL(so)
        L.csr5o_params.argtypes = [C.c_int, C.c_int, C.c_int, _I32P]
        L.csr5o_params.restype = C.c_int
        L.csr5o_tile_ptr.argtypes = [C.c_int] * 5 + [_I32P, _U32P]
        L.csr5o_tile_ptr.restype = None
        L.csr5o_tile_desc.argtypes = [C.c_int] * 7 + [_I32P, _U32P, _U32P, _I32P]
        L.csr5o_tile_desc.restype = C.c_int
        L.csr5o_desc_offset.argtypes = [C.c_int] * 6 + [_I32P, _U32P, _U32P, _I32P, _I32P]
        L.csr5o_desc_offset.restype = None
        L.csr5o_transpose.argtypes = [C.c_int, C.c_int, C.c_int, _U32P, C.c_void_p, C.c_int, C.c_int]
        L.csr5o_transpose.restype = None
        for name, vp in (("csr5o_spmv_f64", _F64P), ("csr5o_spmv_f32", _F32P)):
            f = getattr(L, name)
            f.argtypes = [C.c_int] * 7 + [_I32P, _I32P, vp, _U32P, _U32P, _I32P, _I32P, vp, C.c_int, vp, vp]
            f.restype = None
        L.csr5o_csr_spmv_f64.argtypes = [C.c_int, _I32P, _I32P, _F64P, _F64P, _F64P]
        L.csr5o_csr_spmv_f32.argtypes = [C.c_int, _I32P, _I32P, _F32P, _F32P, _F32P]
        L.csr5o_num_threads.restype = C.c_int
        L.csr5o_mtx_read.argtypes = [C.c_char_p, _I32P]
        L.csr5o_mtx_read.restype = C.c_int
        L.csr5o_mtx_coo.argtypes = [_I32P, _I32P, _F64P]
        L.csr5o_mtx_coo.restype = None
        L.csr5o_mtx_csr.argtypes = [_I32P, _I32P, _F64P]
        L.csr5o_mtx_csr.restype = None
        L.csr5o_coo_nnz.argtypes = [C.c_int, _I32P, _I32P, C.c_int]
        L.csr5o_coo_nnz.restype = C.c_int
        L.csr5o_coo_to_csr.argtypes = [C.c_int, C.c_int, _I32P, _I32P, C.c_void_p, C.c_int, _I32P, _I32P,
                                       C.c_void_p]
        L.csr5o_coo_to_csr.restype = None
        self.L = L

    # ---- Matrix Market ingest (oracle/mtx_oracle.c; reference main.cpp:126-281) ----
    def mtx_read(self, path: str) -> "MtxIngest":
        """Sequential fscanf ingest exactly as the reference CLI does.  Raises MtxExit(code)."""
        dims = np.zeros(6, dtype=np.int32)
        rc = self.L.csr5o_mtx_read(os.fsencode(path), dims)
        if rc:
            raise MtxExit(rc)
        m, n, nnz, nz, symm, fld = (int(v) for v in dims)
        row = np.zeros(max(nz, 1), dtype=np.int32)
        col = np.zeros(max(nz, 1), dtype=np.int32)
        val = np.zeros(max(nz, 1), dtype=np.float64)
        self.L.csr5o_mtx_coo(row, col, val)
        row_ptr = np.zeros(m + 1, dtype=np.int32)
        ci = np.zeros(max(nnz, 1), dtype=np.int32)
        cv = np.zeros(max(nnz, 1), dtype=np.float64)
        self.L.csr5o_mtx_csr(row_ptr, ci, cv)
        return MtxIngest(m, n, nnz, nz, bool(symm), fld, row[:nz], col[:nz], val[:nz], row_ptr, ci[:nnz],
                         cv[:nnz])

    def coo_to_csr(self, m: int, row, col, val, symmetric: bool):
        """The reference's counting scatter (main.cpp:213-275) on COO triplets in file order."""
        row = np.ascontiguousarray(row, dtype=np.int32)
        col = np.ascontiguousarray(col, dtype=np.int32)
        nz = int(row.size)
        r_ = row if nz else np.zeros(1, np.int32)
        c_ = col if nz else np.zeros(1, np.int32)
        nnz = int(self.L.csr5o_coo_nnz(nz, r_, c_, int(symmetric)))
        row_ptr = np.zeros(m + 1, dtype=np.int32)
        ci = np.zeros(max(nnz, 1), dtype=np.int32)
        if val is None:
            self.L.csr5o_coo_to_csr(m, nz, r_, c_, None, int(symmetric), row_ptr, ci, None)
            return row_ptr, ci[:nnz], None
        v = np.ascontiguousarray(val, dtype=np.float64)
        v_ = v if nz else np.zeros(1)
        cv = np.zeros(max(nnz, 1), dtype=np.float64)
        self.L.csr5o_coo_to_csr(m, nz, r_, c_, v_.ctypes.data, int(symmetric), row_ptr, ci, cv.ctypes.data)
        return row_ptr, ci[:nnz], cv[:nnz]

    def num_threads(self) -> int:
        return int(self.L.csr5o_num_threads())

    def params(self, omega: int, sigma: int, nnz: int):
        out = np.zeros(4, dtype=np.int32)
        err = self.L.csr5o_params(omega, sigma, nnz, out)
        if err:
            raise ValueError(f"csr5o_params: error {err}")
        return tuple(int(v) for v in out)

    def convert(self, omega: int, sigma: int, m: int, row_ptr, col, val) -> Csr5Format:
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        col = np.array(col, dtype=np.int32, copy=True)
        val = np.array(val, copy=True)
        assert val.dtype in (np.float64, np.float32)
        nnz = int(row_ptr[m])
        bit_y, bit_ss, num_packet, p = self.params(omega, sigma, nnz)
        tile_ptr = np.zeros(p + 1, dtype=np.uint32)
        tile_desc = np.zeros(max(p * omega * num_packet, 1), dtype=np.uint32)
        offset_ptr = np.zeros(p + 1, dtype=np.int32)
        num_offsets = 0
        tail_start = 0
        if p > 0:
            self.L.csr5o_tile_ptr(omega, sigma, p, m, nnz, row_ptr, tile_ptr)
            tail_start = int(tile_ptr[p - 1] & 0x7FFFFFFF)
            num_offsets = self.L.csr5o_tile_desc(omega, sigma, p, m, bit_y, bit_ss, num_packet,
                                                 row_ptr, tile_ptr, tile_desc, offset_ptr)
        offset = np.full(max(num_offsets, 1), OFFSET_SENTINEL, dtype=np.int32)
        if num_offsets:
            self.L.csr5o_desc_offset(omega, sigma, p, bit_y, bit_ss, num_packet, row_ptr, tile_ptr,
                                     tile_desc, offset_ptr, offset)
        if p > 0:
            self.L.csr5o_transpose(omega, sigma, nnz, tile_ptr, col.ctypes.data, 4, 1)
            self.L.csr5o_transpose(omega, sigma, nnz, tile_ptr, val.ctypes.data, val.itemsize, 1)
        return Csr5Format(omega, sigma, m, nnz, bit_y, bit_ss, num_packet, p, tail_start,
                          num_offsets, tile_ptr, tile_desc[: p * omega * num_packet], offset_ptr,
                          offset[:num_offsets], col, val)

    def revert(self, fmt: Csr5Format):
        """CSR5 -> CSR: inverse tile transpose (anonymouslib_avx2.h:78-102)."""
        col = fmt.col.copy()
        val = fmt.val.copy()
        if fmt.p > 0:
            self.L.csr5o_transpose(fmt.omega, fmt.sigma, fmt.nnz, fmt.tile_ptr, col.ctypes.data, 4, 0)
            self.L.csr5o_transpose(fmt.omega, fmt.sigma, fmt.nnz, fmt.tile_ptr, val.ctypes.data,
                                   val.itemsize, 0)
        return col, val

    def spmv(self, fmt: Csr5Format, row_ptr, x, y0=None) -> np.ndarray:
        vt = fmt.val.dtype
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        x = np.ascontiguousarray(x, dtype=vt)
        y = np.zeros(fmt.m, dtype=vt) if y0 is None else np.array(y0, dtype=vt, copy=True)
        cal = np.zeros(max(fmt.p, 1), dtype=vt)
        f = self.L.csr5o_spmv_f64 if vt == np.float64 else self.L.csr5o_spmv_f32
        offset = fmt.offset if fmt.num_offsets else np.zeros(1, dtype=np.int32)
        desc = fmt.tile_desc if fmt.tile_desc.size else np.zeros(1, dtype=np.uint32)
        f(fmt.omega, fmt.sigma, fmt.p, fmt.m, fmt.bit_y, fmt.bit_ss, fmt.num_packet, row_ptr,
          fmt.col, fmt.val, fmt.tile_ptr, desc, fmt.offset_ptr, offset, cal, fmt.tail_start, x, y)
        return y

    def csr_spmv(self, m, row_ptr, col, val, x) -> np.ndarray:
        val = np.ascontiguousarray(val)
        vt = val.dtype
        y = np.zeros(m, dtype=vt)
        f = self.L.csr5o_csr_spmv_f64 if vt == np.float64 else self.L.csr5o_csr_spmv_f32
        f(m, np.ascontiguousarray(row_ptr, dtype=np.int32), np.ascontiguousarray(col, dtype=np.int32),
          val, np.ascontiguousarray(x, dtype=vt), y)
        return y


class Reference:
    """The reference's own CSR5_avx2 code (prebuilt oracle/_ref/*.so)."""

    @staticmethod
    def available(omega: int | None = None) -> bool:
        names = ["libref_avx2.so"] if omega is None else [f"libref_fmt_w{omega}.so"]
        return all(os.path.exists(os.path.join(_HERE, "_ref", n)) for n in names)

    def __init__(self) -> None:
        self._fmt = {}
        self._avx2 = None

    def _fmt_lib(self, omega: int):
        if omega not in self._fmt:
            L = C.CDLL(os.path.join(_HERE, "_ref", f"libref_fmt_w{omega}.so"))
            assert L.ref_fmt_omega() == omega
            L.ref_fmt_params.argtypes = [C.c_int, C.c_int, _I32P]
            L.ref_fmt_tile.argtypes = [C.c_int] * 7 + [_I32P, _U32P, _U32P, _I32P]
            L.ref_fmt_tile.restype = C.c_int
            L.ref_fmt_offset.argtypes = [C.c_int] * 5 + [_I32P, _U32P, _U32P, _I32P, _I32P]
            L.ref_fmt_offset.restype = None
            L.ref_fmt_transpose_f64.argtypes = [C.c_int, C.c_int, _U32P, _I32P, _F64P, C.c_int]
            L.ref_fmt_transpose_f32.argtypes = [C.c_int, C.c_int, _U32P, _I32P, _F32P, C.c_int]
            self._fmt[omega] = L
        return self._fmt[omega]

    def convert(self, omega: int, sigma: int, m: int, row_ptr, col, val) -> Csr5Format:
        L = self._fmt_lib(omega)
        rp = np.zeros(m + 2, dtype=np.int32)
        rp[: m + 1] = row_ptr
        rp[m + 1] = 0x7FFFFFFF  # the reference reads one past the end for the last tile
        col = np.array(col, dtype=np.int32, copy=True)
        val = np.array(val, copy=True)
        nnz = int(rp[m])
        out = np.zeros(4, dtype=np.int32)
        err = L.ref_fmt_params(sigma, nnz, out)
        if err:
            raise ValueError(f"ref_fmt_params: error {err}")
        bit_y, bit_ss, num_packet, p = (int(v) for v in out)
        tile_ptr = np.zeros(p + 1, dtype=np.uint32)
        tile_desc = np.zeros(max(p * omega * num_packet, 1), dtype=np.uint32)
        offset_ptr = np.zeros(p + 1, dtype=np.int32)
        num_offsets = L.ref_fmt_tile(sigma, p, m, nnz, bit_y, bit_ss, num_packet, rp, tile_ptr,
                                     tile_desc, offset_ptr)
        offset = np.full(max(num_offsets, 1), OFFSET_SENTINEL, dtype=np.int32)
        if num_offsets:
            L.ref_fmt_offset(sigma, p, bit_y, bit_ss, num_packet, rp, tile_ptr, tile_desc,
                             offset_ptr, offset)
        if val.dtype == np.float64:
            L.ref_fmt_transpose_f64(sigma, nnz, tile_ptr, col, val, 1)
        else:
            L.ref_fmt_transpose_f32(sigma, nnz, tile_ptr, col, val, 1)
        tail_start = int(tile_ptr[p - 1] & 0x7FFFFFFF)
        return Csr5Format(omega, sigma, m, nnz, bit_y, bit_ss, num_packet, p, tail_start,
                          int(num_offsets), tile_ptr, tile_desc[: p * omega * num_packet],
                          offset_ptr, offset[:num_offsets], col, val)

    @staticmethod
    def ingest_available() -> bool:
        return os.path.exists(os.path.join(_HERE, "_ref", "libref_ingest.so"))

    def ingest(self, path: str):
        """CSR exactly as the reference CLI holds it at main.cpp:281 -> (m, n, row_ptr, col, val)."""
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_ingest.so"))
        L.ref_ingest_run.argtypes = [C.c_char_p, _I32P]
        L.ref_ingest_run.restype = C.c_int
        L.ref_ingest_fetch.argtypes = [_I32P, _I32P, _F64P]
        L.ref_ingest_fetch.restype = None
        dims = np.zeros(3, dtype=np.int32)
        rc = L.ref_ingest_run(os.fsencode(path), dims)
        if rc:
            raise MtxExit(rc)
        m, n, nnz = (int(v) for v in dims)
        row_ptr = np.zeros(m + 1, dtype=np.int32)
        col = np.zeros(max(nnz, 1), dtype=np.int32)
        val = np.zeros(max(nnz, 1), dtype=np.float64)
        L.ref_ingest_fetch(row_ptr, col, val)
        return m, n, row_ptr, col[:nnz], val[:nnz]

    def cli(self, path: str) -> tuple:
        """Run the reference's whole CLI (CSR5_avx2/main.cpp) on `path` -> (exit code, stdout text)."""
        L = C.CDLL(os.path.join(_HERE, "_ref", "libref_ingest.so"))
        L.ref_cli_run.argtypes = [C.c_char_p, C.c_char_p, C.c_int]
        L.ref_cli_run.restype = C.c_int
        buf = C.create_string_buffer(1 << 16)
        rc = L.ref_cli_run(os.fsencode(path), buf, len(buf))
        return rc, buf.value.decode(errors="replace")

    def _avx2_lib(self):
        if self._avx2 is None:
            L = C.CDLL(os.path.join(_HERE, "_ref", "libref_avx2.so"))
            L.ref_avx2_spmv.argtypes = [C.c_int, C.c_int, C.c_int, _I32P, _I32P, _F64P, _F64P, _F64P,
                                        C.c_int, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]
            L.ref_avx2_spmv.restype = C.c_int
            self._avx2 = L
        return self._avx2

    def avx2_threads(self) -> int:
        return int(self._avx2_lib().ref_avx2_threads())

    def avx2_set_threads(self, n: int) -> None:
        """ambient OpenMP thread count of the reference build (omp_set_num_threads)"""
        self._avx2_lib().ref_avx2_set_threads(int(n))

    def avx2_spmv(self, m, n, row_ptr, col, val, x, y0=None, warm=0, runs=0):
        """y from CSR5_avx2 (omega=4, sigma=16, fp64).  Returns (y, ms_per_run, convert_ms)."""
        L = self._avx2_lib()
        row_ptr = np.ascontiguousarray(row_ptr, dtype=np.int32)
        nnz = int(row_ptr[m])
        y = np.zeros(m, dtype=np.float64) if y0 is None else np.array(y0, dtype=np.float64, copy=True)
        ms = C.c_double(0.0)
        cms = C.c_double(0.0)
        err = L.ref_avx2_spmv(m, n, nnz, row_ptr, np.ascontiguousarray(col, dtype=np.int32),
                              np.ascontiguousarray(val, dtype=np.float64),
                              np.ascontiguousarray(x, dtype=np.float64), y, warm, runs,
                              C.byref(ms), C.byref(cms))
        if err:
            raise RuntimeError(f"reference CSR5_avx2 returned {err}")
        return y, ms.value, cms.value

"""CPU tests (no GPU): pin the oracle (oracle/csr5_oracle.c) against the reference.

1. KAT-0: the survey's worked known-answer, typed in by hand.
2. Committed golden vectors (tests/golden/*.npz) produced by the reference's own CSR5_avx2 code
   (oracle/gen_golden.py): format arrays bit-exact at omega 64/32/4, y of the real AVX2 SpMV.
3. When oracle/_ref/*.so is present (build container, or shipped prebuilt to the GPU box): the
   oracle against the live reference on the whole matrix zoo and more sigmas.
Comparison rules are those of SURVEY.md section 8(c).
"""
import glob
import os

import numpy as np
import pytest

from benchmark_spmv_using_csr5_amd import matrices as M
from oracle.csr5_oracle import Oracle, Reference
from tests import zoo

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))
FORMATS = [(64, 4), (64, 16), (64, 24), (32, 8), (4, 16)]


def assert_format_equal(a, b, tag=""):
    """a, b: Csr5Format.  bit 31 of tile_ptr[p-1] is masked (the AVX2 code reads past row_ptr there);
    tile_desc is compared on tiles 0..p-2; offset only where the reference wrote a slot."""
    assert (a.bit_y, a.bit_ss, a.num_packet, a.p, a.num_offsets, a.tail_start) == \
        (b.bit_y, b.bit_ss, b.num_packet, b.p, b.num_offsets, b.tail_start), tag
    p = a.p
    ta, tb = a.tile_ptr.copy(), b.tile_ptr.copy()
    ta[p - 1] &= 0x7FFFFFFF
    tb[p - 1] &= 0x7FFFFFFF
    assert np.array_equal(ta, tb), (tag, "tile_ptr")
    n = (p - 1) * a.omega * a.num_packet
    assert np.array_equal(a.tile_desc[:n], b.tile_desc[:n]), (tag, "tile_desc")
    assert np.array_equal(a.offset_ptr, b.offset_ptr), (tag, "offset_ptr")
    written = b.offset != -1
    assert np.array_equal(a.offset[written], b.offset[written]), (tag, "offset")
    assert np.array_equal(a.col, b.col), (tag, "transposed column_index")
    assert np.array_equal(a.val, b.val), (tag, "transposed value")


def test_kat0_known_answer(oracle):
    """omega=4, sigma=4, row lengths 3,0,5,1,9,0,0,14,2 (values typed from there)."""
    mat = zoo.kat0()
    assert mat.row_ptr.tolist() == [0, 3, 3, 8, 9, 18, 18, 18, 32, 34]
    val = np.arange(mat.nnz, dtype=np.float64)
    f = oracle.convert(4, 4, mat.m, mat.row_ptr, mat.col, val)
    assert (f.bit_y, f.bit_ss, f.num_packet, f.p) == (4, 2, 1, 3)
    assert f.tile_ptr.tolist() == [0x80000000, 0x80000004, 0x00000008, 0x00000009]
    assert f.tile_desc.tolist() == [0x06400000, 0x10000000, 0x17000000, 0x30000000,
                                    0x0c800000, 0x10000000, 0x10000000, 0x10000000, 0, 0, 0, 0]
    assert f.offset_ptr.tolist() == [0, 4, 6, 6]
    assert f.offset.tolist() == [1, 2, 3, -1, 2, -1]  # -1 = slot the reference never writes
    assert f.val.astype(int).tolist() == [0, 4, 8, 12, 1, 5, 9, 13, 2, 6, 10, 14, 3, 7, 11, 15,
                                          16, 20, 24, 28, 17, 21, 25, 29, 18, 22, 26, 30, 19, 23, 27, 31,
                                          32, 33]
    col, v = oracle.revert(f)
    assert np.array_equal(v, val) and np.array_equal(col, mat.col)


def _golden_format(g, omega, sigma, dtype_val="val_int"):
    from oracle.csr5_oracle import Csr5Format
    k = f"fmt_w{omega}_s{sigma}_"
    bit_y, bit_ss, num_packet, p, num_offsets, tail_start = (int(v) for v in g[k + "params"])
    return Csr5Format(omega, sigma, int(g["m"]), int(g["row_ptr"][-1]), bit_y, bit_ss, num_packet, p,
                      tail_start, num_offsets, g[k + "tile_ptr"], g[k + "tile_desc"],
                      g[k + "offset_ptr"], g[k + "offset"], g[k + "col"], g[k + "val"])


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_matches_reference_goldens(oracle, path):
    g = np.load(path)
    m, n = int(g["m"]), int(g["n"])
    row_ptr, col = g["row_ptr"], g["col"]
    nonempty = np.diff(row_ptr) > 0
    for omega, sigma in FORMATS:
        ours = oracle.convert(omega, sigma, m, row_ptr, col, g["val_int"])
        assert_format_equal(ours, _golden_format(g, omega, sigma), (os.path.basename(path), omega, sigma))
    # y of the real CSR5_avx2 SpMV (omega 4, sigma 16): exact on the CLI's integer data ...
    f = oracle.convert(4, 16, m, row_ptr, col, g["val_int"])
    y = oracle.spmv(f, row_ptr, g["x_int"], y0=np.full(m, 777.0))
    assert np.array_equal(y[nonempty], g["y_avx2_int"][nonempty])
    # ... and within 1e-12 of sum|a x| on uniform(-1,1) data (summation order differs: FMA, hscan)
    f = oracle.convert(4, 16, m, row_ptr, col, g["val_real"])
    y = oracle.spmv(f, row_ptr, g["x_real"], y0=np.full(m, 777.0))
    scale = oracle.csr_spmv(m, row_ptr, col, np.abs(g["val_real"]), np.abs(g["x_real"]))
    assert np.all(np.abs(y - g["y_avx2_real"])[nonempty] <= 1e-12 * np.maximum(scale[nonempty], 1.0))
    # rows the reference leaves untouched (empty, before the tail) keep the caller's value in both
    untouched = (~nonempty) & (np.arange(m) < f.tail_start)
    assert np.all(y[untouched] == 777.0) and np.all(g["y_avx2_real"][untouched] == 777.0)


@pytest.mark.parametrize("omega", [4, 32, 64])
def test_oracle_spmv_all_omegas_on_zoo(oracle, omega):
    """The oracle's SpMV against the reference CLI's own check, the scalar CSR loop (exact on integer
    data), for every omega/sigma the GPU tests use, plus the CSR5 -> CSR round trip."""
    for mat in zoo.small_zoo():
        val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=5, mode="int")
        ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
        nonempty = np.diff(mat.row_ptr) > 0
        for sigma in (1, 3, 4, 7, 16, 17, 32):
            f = oracle.convert(omega, sigma, mat.m, mat.row_ptr, mat.col, val)
            y = oracle.spmv(f, mat.row_ptr, x, y0=np.full(mat.m, 777.0))
            assert np.array_equal(y[nonempty], ref[nonempty]), (mat.name, omega, sigma)
            untouched = (~nonempty) & (np.arange(mat.m) < f.tail_start)
            assert np.all(y[untouched] == 777.0)
            assert np.all(y[(~nonempty) & (np.arange(mat.m) >= f.tail_start)] == 0.0)  # tail rows are written
            c, v = oracle.revert(f)
            assert np.array_equal(c, mat.col) and np.array_equal(v, val)


def test_oracle_fp32_and_unsupported_omega(oracle):
    mat = zoo.small_zoo()[1]
    val, x = M.fill_values(mat.nnz, mat.n, np.float32, seed=5, mode="int")
    f = oracle.convert(64, 16, mat.m, mat.row_ptr, mat.col, val)
    y = oracle.spmv(f, mat.row_ptr, x)
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    nonempty = np.diff(mat.row_ptr) > 0
    assert y.dtype == np.float32 and np.array_equal(y[nonempty], ref[nonempty])
    with pytest.raises(ValueError):  # bit_y + bit_ss > 31 -> ANONYMOUSLIB_UNSUPPORTED_CSR5_OMEGA
        oracle.params(1 << 16, 1 << 10, 100)


@pytest.mark.skipif(not (Reference.available() and Reference.available(64)), reason="oracle/_ref not built")
def test_oracle_against_live_reference_on_zoo(oracle):
    ref = Reference()
    for mat in zoo.small_zoo():
        val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=21, mode="int")
        for omega in (4, 32, 64):
            for sigma in (4, 5, 12, 16, 17, 32):
                assert_format_equal(oracle.convert(omega, sigma, mat.m, mat.row_ptr, mat.col, val),
                                    ref.convert(omega, sigma, mat.m, mat.row_ptr, mat.col, val),
                                    (mat.name, omega, sigma))
        y_ref, _, _ = ref.avx2_spmv(mat.m, mat.n, mat.row_ptr, mat.col, val, x, y0=np.full(mat.m, 777.0))
        f = oracle.convert(4, 16, mat.m, mat.row_ptr, mat.col, val)
        y = oracle.spmv(f, mat.row_ptr, x, y0=np.full(mat.m, 777.0))
        nonempty = np.diff(mat.row_ptr) > 0
        assert np.array_equal(y[nonempty], y_ref[nonempty]), mat.name


def test_y_offset_recomputed_from_bit_flags(oracle):
    """The kernels that read no descriptor word (k_spmv_range; k_spmv with the narrow column codes, whose bit 15 carries the
    row-start flag) recompute the descriptor's y_offset from the lane's bit flags: segments that start in the lane, exclusive
    wave prefix, minus one for lanes > 0 (csr5_spmv.hip, csr5_hot.hip; reference field: format_cuda.h:161-267).  The two must
    agree on every lane that owns a flag -- the only lanes that use it -- with and without empty rows, one and two packets."""
    rng = np.random.default_rng(1)
    mats = zoo.small_zoo() + [M.nd24k_like(scale=0.03, dtype=np.float64)]
    for k in range(4):
        lens = rng.integers(0, 60, size=3000)
        lens[rng.random(3000) < 0.4] = 0
        mats.append(M.csr_from_row_lengths(lens, 5000, rng, band=0.3, name=f"r{k}"))
    checked = 0
    for mat in mats:
        for sigma in (8, 12, 16, 24, 32):
            f = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, np.ones(mat.nnz))
            if f.p < 2:
                continue
            bit_all = f.bit_y + f.bit_ss
            d = f.tile_desc.reshape(f.p, f.num_packet, 64).astype(np.uint64)
            for t in range(f.p - 1):
                flags = (d[t, 0] << np.uint64(bit_all)) & np.uint64(0xFFFFFFFF)
                if f.num_packet > 1:
                    flags |= d[t, 1] >> np.uint64(32 - bit_all)
                stored = (d[t, 0] >> np.uint64(32 - f.bit_y)).astype(np.int64)
                bits = (flags[:, None] >> (np.uint64(31) - np.arange(sigma, dtype=np.uint64))[None, :]) & np.uint64(1)
                f0 = bits[:, 0].astype(bool)
                f0[0] = True
                stop = bits[:, 1:].sum(axis=1).astype(np.int64)
                present = f0 | (stop > 0)
                segn = np.maximum(stop - np.where(f0, 0, 1) + np.where(present, 1, 0), 0)
                y = np.cumsum(segn) - segn - 1
                y[0] = 0
                assert np.array_equal(y[present], stored[present]), (mat.name, sigma, t)
                checked += 1
    assert checked > 1000

"""GPU parity tests of the range-walking, software-pipelined tile kernel (csr5_walk.hip, CSR5HIP_OPT_TILE_WALK) against the
CPU oracle, through the C ABI.

What it must compute is the reference's tile kernel + calibrate + tail (CSR5_cuda/detail/cuda/csr5_spmv_cuda.h:59-419) on the
unchanged format arrays.  Bars: y bit-exact on the reference CLI's integer data for every way of cutting the tiles into ranges
(1 range ... one range per tile), within 1e-12 * sum|a x| (fp64) / 1e-5 (fp32) on real data, bit-reproducible run to run.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from benchmark_spmv_using_csr5_amd import matrices as M  # noqa: E402
from benchmark_spmv_using_csr5_amd import handle as H  # noqa: E402
from tests import zoo  # noqa: E402
from tests.test_gpu_parity import _run, _check_format, _expected_y, Y_POISON  # noqa: E402

RANGES = [1, 2, 3, 5, 64, 0, 16384]  # 0 = default; more than p - 1 = one range per tile


@pytest.mark.parametrize("sigma", [4, 5, 7, 12, 16])
def test_walk_zoo_integer_data_bit_exact(oracle, sigma):
    """Every zoo matrix (empty rows, hub rows, rows on tile edges, p = 1 ...) x every range count: format untouched, y exact."""
    for mat in zoo.small_zoo():
        val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=5, mode="int")
        fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
        exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
        for ranges in RANGES:
            info = {}
            arrays, col_t, val_t, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, walk=2, walk_ranges=ranges, slabs=0, repeat=2,
                                            info_out=info)
            _check_format(arrays, col_t, val_t, fmt)
            assert info["tile_walk"] == (1 if fmt.p > 1 else 0), (mat.name, sigma, info)
            if fmt.p > 1:
                assert info["walk_ranges"] == min(fmt.p - 1, ranges if ranges else 2048)
            for y in ys:
                assert np.array_equal(y, exp), (mat.name, sigma, ranges, np.flatnonzero(y != exp)[:8])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_walk_x_window_variant(oracle, dtype):
    """x-window variant (forced; sigma 4, 8, 12, 16) on matrices with and without column locality, non-temporal streams on and
    off, with the 16-bit column codes (sigma 8, 12, 16) and without: integer data exact, real data within tolerance and
    bit-reproducible."""
    tol = 1e-12 if dtype == np.float64 else 1e-5
    mats = zoo.small_zoo() + [M.nd24k_like(scale=0.03, dtype=np.float64)]
    for mat in mats:
        for sigma in (4, 8, 12, 16):
            for nt, ranges, narrow_cols in ((0, 3, None), (2, 0, None), (0, 5, 0)):
                val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=30, mode="int")
                if dtype == np.float32:  # keep every partial sum below 2^24
                    val, x = (val % 3).astype(np.float32), (x % 3).astype(np.float32)
                fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
                info = {}
                _, _, _, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, xwin=2, nt=nt, walk=2, walk_ranges=ranges,
                                   slabs=0, info_out=info, narrow_cols=narrow_cols)
                assert info["tile_walk"] == (1 if fmt.p > 1 else 0) and info["walk_x_window"] == info["tile_walk"]
                # 16-bit column codes (every tile of these matrices spans < 32 768 columns): the sigmas k_col16 serves, unless off
                assert info["narrow_columns"] == (1 if fmt.p > 1 and sigma in (8, 12, 16) and narrow_cols is None else 0), info
                exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
                assert np.array_equal(ys[0], exp), (mat.name, sigma, nt, ranges, np.flatnonzero(ys[0] != exp)[:8])
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=31, mode="real")
            fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
            _, _, _, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, xwin=2, walk=2, walk_ranges=7, slabs=0, repeat=2)
            exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
            scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val), np.abs(x))
            assert np.all(np.abs(ys[0] - exp) <= tol * np.maximum(scale, 1.0)), (mat.name, sigma)
            assert np.array_equal(ys[0], ys[1])


def test_walk_real_data_and_fp32(oracle):
    for mat in zoo.small_zoo():
        for sigma, dtype in ((4, np.float64), (6, np.float64), (16, np.float64), (8, np.float32), (16, np.float32)):
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=9, mode="real")
            fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
            exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
            scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val), np.abs(x))
            tol = 1e-12 if dtype == np.float64 else 1e-5
            for ranges in (2, 0):
                _, _, _, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, walk=2, walk_ranges=ranges, slabs=0, repeat=3)
                assert np.all(np.abs(ys[0] - exp) <= tol * np.maximum(scale, 1.0)), (mat.name, sigma, ranges)
                assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2]), "bit-reproducible, handle re-arms itself"


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_walk_rows_spanning_many_ranges(oracle, dtype):
    """Rows that span 2, a few, and more than 64 ranges (the parked-partial path + k_calibrate on ranges), one range per tile and
    a handful of tiles per range; every row of `long-rows` is cut by several range seams on all 8 XCDs."""
    rng = np.random.default_rng(77)
    lens = rng.integers(1500, 9000, size=260)
    lens[::7] = rng.integers(1, 40, size=lens[::7].size)
    mats = [M.csr_from_row_lengths(lens, 50_000, rng, band=0.0, name="long-rows")]
    for k, l in enumerate([[3, 0, 70 * 256 + 17, 5, 1, 0, 2], [1] * 50 + [66 * 1024] + [2] * 30 + [65 * 1024 + 1, 0, 7],
                           [2, 3_000_000, 1, 0, 4]]):
        mats.append(M.csr_from_row_lengths(np.asarray(l), 50000, np.random.default_rng(100 + k), band=0.0, name=f"longrun{k}"))
    for mat in mats:
        for sigma, ranges in ((4, 16384), (4, 300), (16, 0), (16, 16384), (8, 1000)):
            for fill in ("int", "real"):
                val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=3, mode=fill)
                if dtype == np.float32 and fill == "int":
                    val, x = (val % 2).astype(np.float32), (x % 2).astype(np.float32)
                exp = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val.astype(np.float64), x.astype(np.float64))
                scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val).astype(np.float64), np.abs(x).astype(np.float64))
                _, _, _, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, y0=0.0, repeat=3, slabs=0, walk=2,
                                   walk_ranges=ranges)
                assert np.array_equal(ys[0], ys[1]) and np.array_equal(ys[0], ys[2]), (mat.name, sigma, ranges, fill)
                nonempty = np.diff(mat.row_ptr) > 0
                got = ys[0].astype(np.float64)
                if fill == "int" and (dtype == np.float64 or mat.nnz < 2 ** 23):
                    assert np.array_equal(got[nonempty], exp[nonempty]), (mat.name, sigma, ranges)
                else:
                    tol = (1e-12 if dtype == np.float64 else 2e-5) * np.maximum(scale, 1.0)
                    assert np.all(np.abs(got - exp)[nonempty] <= tol[nonempty]), (mat.name, sigma, ranges, fill)


@pytest.mark.parametrize("workload", ["scircuit", "webbase", "nd24k", "rmat20"])
def test_walk_full_size_workloads(oracle, workload):
    """BASELINE config sizes (and an R-MAT 20 with its empty rows) through the walking kernel at the library's defaults for
    sigma / x-window / ranges: exact on integer data against the scalar CSR loop, empty rows untouched."""
    dtype = np.float32 if workload == "nd24k" else np.float64
    mat = {"scircuit": M.scircuit_like, "webbase": M.webbase_like, "nd24k": lambda: M.nd24k_like(dtype=np.float32),
           "rmat20": lambda: M.rmat(20, 16, seed=4)}[workload]()
    val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=12, mode="int")
    if dtype == np.float32:
        val, x = (val % 3).astype(np.float32), (x % 3).astype(np.float32)
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val.astype(np.float64), x.astype(np.float64))
    nonempty = np.diff(mat.row_ptr) > 0
    for ranges in (0, 4096):
        info = {}
        arrays, _, _, ys = _run(mat, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, H.SPMV_FUSED, dtype=dtype, walk=2,
                                walk_ranges=ranges, slabs=0, info_out=info, repeat=2)
        assert info["tile_walk"] == 1, info
        assert info["walk_x_window"] == (1 if workload == "nd24k" else 0), info  # auto: the banded stand-in gets windows
        before_tail = np.arange(mat.m) < arrays["tail_start"]  # (every row of the CSR tail is written, csr5hip.h spmv)
        for y in ys:
            assert np.array_equal(y.astype(np.float64)[nonempty], ref[nonempty]), (workload, ranges)
            assert np.all(y[~nonempty & before_tail] == Y_POISON) and np.all(y[~nonempty & ~before_tail] == 0)

"""GPU parity, ONE hop: the HIP path against the committed golden vectors themselves (tests/golden/*.npz, written by the
reference's own CSR5_avx2 code through oracle/gen_golden.py) -- not through the oracle.

* format arrays at omega = 64, sigma 4 / 16 / 24 (reference format_avx2.h re-instantiated at omega 64): tile_ptr,
  tile_desc, offset_ptr, offset, tile-transposed column_index / value bit-exact (comparison rules of SURVEY.md 8c);
* y of the REAL CSR5_avx2 handle (omega 4, sigma 16): exact on the CLI's integer data whatever our sigma / mode / slab
  options are, and within 1e-12 * sum|a x| (and 1e-6 relative where a row is well conditioned) on uniform(-1, 1) data.
"""
import glob
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from benchmark_spmv_using_csr5_amd import handle as H  # noqa: E402
from benchmark_spmv_using_csr5_amd import matrices as M  # noqa: E402
from tests.test_gpu_parity import Y_POISON, _check_format, _run  # noqa: E402
from tests.test_oracle_golden import _golden_format  # noqa: E402

GOLDEN = sorted(glob.glob(os.path.join(os.path.dirname(__file__), "golden", "*.npz")))


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_against_the_reference_goldens(path):
    g = np.load(path)
    m, n = int(g["m"]), int(g["n"])
    mat = M.CsrMatrix(m, n, g["row_ptr"], g["col"], None, os.path.basename(path)[:-4])
    nonempty = np.diff(mat.row_ptr) > 0
    abs_ax = np.zeros(m)
    np.add.at(abs_ax, np.repeat(np.arange(m), np.diff(mat.row_ptr)), np.abs(g["val_real"]) * np.abs(g["x_real"][mat.col]))
    for sigma in (4, 16, 24):
        gold = _golden_format(g, 64, sigma)
        for mode in (H.SPMV_TWO_PASS, H.SPMV_FUSED):
            arrays, col_t, val_t, ys = _run(mat, g["val_int"], g["x_int"], sigma, mode)
            _check_format(arrays, col_t, val_t, gold)
            assert np.array_equal(ys[0][nonempty], g["y_avx2_int"][nonempty]), (sigma, mode)
            untouched = (~nonempty) & (np.arange(m) < gold.tail_start)
            assert np.all(ys[0][untouched] == Y_POISON), "rows the reference leaves untouched stay untouched"
            _, _, _, yr = _run(mat, g["val_real"], g["x_real"], sigma, mode)
            err = np.abs(yr[0] - g["y_avx2_real"])
            assert np.all(err[nonempty] <= 1e-12 * np.maximum(abs_ax[nonempty], 1.0)), (sigma, mode)
            well = nonempty & (np.abs(g["y_avx2_real"]) >= 1e-3 * abs_ax)
            assert np.all(err[well] <= 1e-6 * np.abs(g["y_avx2_real"][well])), (sigma, mode)
    if mat.nnz >= 2 * 64 * 4:
        # the kernel-side column streams (round 6: column words with the row-start flag in bit 31, forced; round 5: narrow codes
        # with the x-window forced) against the reference's arrays and y directly
        gold = _golden_format(g, 64, 4)
        for kw in (dict(flagged=2, xwin=0), dict(flagged=2, xwin=0, defer=2), dict(xwin=2)):
            arrays, col_t, val_t, ys = _run(mat, g["val_int"], g["x_int"], 4 if "flagged" in kw else 16, H.SPMV_FUSED, slabs=0, **kw)
            _check_format(arrays, col_t, val_t, gold if "flagged" in kw else _golden_format(g, 64, 16))
            assert np.array_equal(ys[0][nonempty], g["y_avx2_int"][nonempty]), kw
        # the kernel-side structures (column slabs, forced; hot table where the slab count allows) change nothing exposed
        gold = _golden_format(g, 64, 4)
        arrays, col_t, val_t, ys = _run(mat, g["val_int"], g["x_int"], 4, H.SPMV_FUSED, slabs=8, hot=2)
        _check_format(arrays, col_t, val_t, gold)
        assert np.array_equal(ys[0][nonempty], g["y_avx2_int"][nonempty])

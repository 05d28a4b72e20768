"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star): tile_ptr / tile_desc / offsets BIT-EXACT against the reference format
algorithm at omega = 64; y bit-exact on the reference CLI's integer data, and within 1e-6 relative
for fp64 on real data (the tolerance is written in each test).
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from benchmark_spmv_using_csr5_amd import matrices as M  # noqa: E402
from benchmark_spmv_using_csr5_amd import handle as H  # noqa: E402
from benchmark_spmv_using_csr5_amd import _capi  # noqa: E402
from tests import zoo  # noqa: E402

DEV = "cuda:0"
SIGMAS = [4, 5, 7, 12, 16, 17, 24, 32]
Y_POISON = 777.0


def _device_csr(mat, val, dtype):
    rp = torch.from_numpy(mat.row_ptr.astype(np.int32)).to(DEV)
    ci = torch.from_numpy(mat.col.astype(np.int32)).to(DEV)
    va = torch.from_numpy(val.astype(dtype)).to(DEV)
    return rp, ci, va


def _run(mat, val, x, sigma, mode, dtype=np.float64, y0=Y_POISON, repeat=1, xwin=None, ldsy=None, nt=None,
         slabs=None, slab_shift=None, zero_empty=None, info_out=None, hot=None, x_snapshot=None, narrow=None,
         narrow_cols=None, defer=None, x_misaligned=False, flagged=None):
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    rp, ci, va = _device_csr(mat, val, dtype)
    xd = torch.from_numpy(x.astype(dtype)).to(DEV)
    if x_misaligned:  # a caller's x need only be element-aligned: one element into a larger allocation
        xd = torch.cat([torch.zeros(1, dtype=tdt, device=DEV), xd])[1:]
        assert xd.data_ptr() % 16 != 0
    yd = torch.full((mat.m,), y0, dtype=tdt, device=DEV)
    A = H.anonymouslibHandle(mat.m, mat.n, dtype=np.dtype(dtype).name)
    assert A.inputCSR(mat.nnz, rp, ci, va) == 0
    assert A.setX(xd) == 0
    assert A.setSigma(sigma) == 0
    assert A.setSpmvMode(mode) == 0
    if xwin is not None:
        assert A.setXWindow(xwin) == 0
    if ldsy is not None:
        assert A.setLdsY(ldsy) == 0
    if nt is not None:
        assert A.setStreamNT(nt) == 0
    if slabs is not None:
        assert A.setColumnSlabs(slabs) == 0
    if slab_shift is not None:
        assert A.setSlabShift(slab_shift) == 0
    if zero_empty is not None:
        assert A.setZeroEmptyRows(zero_empty) == 0
    if hot is not None:
        assert A.setSlabHot(hot) == 0
    if x_snapshot is not None:
        assert A.setXSnapshot(x_snapshot) == 0
    if narrow is not None:
        assert A.setNarrowValues(narrow) == 0
    if narrow_cols is not None:
        assert A.setNarrowColumns(narrow_cols) == 0
    if defer is not None:
        assert A.setDeferCarries(defer) == 0
    if flagged is not None:
        assert A.setFlaggedColumns(flagged) == 0
    assert A.spmv(1.0, yd) == H.ANONYMOUSLIB_UNSUPPORTED_CSR_SPMV  # still CSR (anonymouslib_cuda.h:268-271)
    assert A.asCSR5() == 0, _capi.last_error()
    arrays = A.csr5_arrays()
    if info_out is not None:
        i = A.info()
        info_out.update(column_slabs=i.column_slabs, slab_segments=i.slab_segments, slab_sigma=i.slab_sigma,
                        slab_tiles=i.slab_tiles, sigma=i.sigma, slab_hot=i.slab_hot,
                        slab_hot_cover_pct=i.slab_hot_cover_pct, p=i.p, x_window_active=i.x_window_active,
                        narrow_columns=i.narrow_columns, carries_deferred=i.carries_deferred,
                        flagged_columns=i.flagged_columns)
    col_t = ci.cpu().numpy().copy()
    val_t = va.cpu().numpy().copy()
    ys = []
    for _ in range(repeat):
        yd.fill_(y0)
        assert A.spmv(1.0, yd) == 0
        torch.cuda.synchronize()
        ys.append(yd.cpu().numpy().copy())
    assert A.destroy() == 0
    torch.cuda.synchronize()
    # destroy() == asCSR(): the caller's arrays are back in CSR order (anonymouslib_cuda.h:78-102)
    assert np.array_equal(ci.cpu().numpy(), mat.col)
    assert np.array_equal(va.cpu().numpy(), val.astype(dtype))
    A.close()
    return arrays, col_t, val_t, ys


def _check_format(arrays, col_t, val_t, fmt):
    """Comparison rules of SURVEY.md section 8(c)."""
    p = fmt.p
    assert (arrays["sigma"], arrays["bit_y"], arrays["bit_ss"], arrays["num_packet"], arrays["p"]) == \
        (fmt.sigma, fmt.bit_y, fmt.bit_ss, fmt.num_packet, fmt.p)
    if p == 0:
        return
    assert arrays["tail_start"] == fmt.tail_start
    assert arrays["num_offsets"] == fmt.num_offsets
    a, b = arrays["tile_ptr"].copy(), fmt.tile_ptr.copy()
    a[p - 1] &= 0x7FFFFFFF  # bit 31 of the last entry: the AVX2 oracle reads past row_ptr there
    b[p - 1] &= 0x7FFFFFFF
    assert np.array_equal(a, b), "tile_ptr"
    n = (p - 1) * fmt.omega * fmt.num_packet
    assert np.array_equal(arrays["tile_desc"][:n], fmt.tile_desc[:n]), "tile_desc"
    assert np.array_equal(arrays["offset_ptr"], fmt.offset_ptr), "offset_ptr"
    written = fmt.offset != -1
    assert np.array_equal(arrays["offset"][written], fmt.offset[written]), "offset"
    assert np.array_equal(col_t, fmt.col), "transposed column_index"
    assert np.array_equal(val_t, fmt.val), "transposed value"


def _expected_y(oracle, fmt, mat, x, y0):
    return oracle.spmv(fmt, mat.row_ptr, x, y0=np.full(mat.m, y0, dtype=fmt.val.dtype))


@pytest.mark.parametrize("mode", [H.SPMV_TWO_PASS, H.SPMV_FUSED])
@pytest.mark.parametrize("sigma", SIGMAS)
def test_zoo_integer_data_bit_exact(oracle, sigma, mode):
    """Reference CLI data (rand()%10, main.cu:336-347): every partial sum is exact, so format AND y
    must be bit-identical to the oracle, including which rows are left untouched."""
    for mat in zoo.small_zoo():
        val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=5, mode="int")
        fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
        arrays, col_t, val_t, ys = _run(mat, val, x, sigma, mode)
        _check_format(arrays, col_t, val_t, fmt)
        exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
        assert np.array_equal(ys[0], exp), (mat.name, sigma, mode, np.flatnonzero(ys[0] != exp)[:8])
        # and against the reference CLI's own check, the scalar CSR loop (main.cu:351-363)
        ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
        nonempty = np.diff(mat.row_ptr) > 0
        assert np.array_equal(ys[0][nonempty], ref[nonempty]), (mat.name, sigma)


@pytest.mark.parametrize("mode", [H.SPMV_TWO_PASS, H.SPMV_FUSED])
@pytest.mark.parametrize("sigma", [4, 16, 24])
def test_real_data_fp64_tolerance(oracle, sigma, mode):
    """fp64 on real-valued data: |y - y_oracle| <= 1e-6 * |y_oracle| on positive data (no
    cancellation), and <= 1e-12 * sum|a_ij x_j| on signed data."""
    for mat in zoo.small_zoo():
        for fill in ("pos", "real"):
            val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=9, mode=fill)
            fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
            _, _, _, ys = _run(mat, val, x, sigma, mode, repeat=2)
            exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
            scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val), np.abs(x))
            for y in ys:  # second call: the handle re-arms itself, y need not be zeroed
                if fill == "pos":
                    assert np.all(np.abs(y - exp) <= 1e-6 * np.abs(exp)), (mat.name, sigma, mode)
                assert np.all(np.abs(y - exp) <= 1e-12 * np.maximum(scale, 1.0)), (mat.name, sigma, mode)
            assert np.array_equal(ys[0], ys[1]), "both SpMV modes are bit-reproducible run to run"


@pytest.mark.parametrize("mode", [H.SPMV_TWO_PASS, H.SPMV_FUSED])
def test_fp32_path(oracle, mode):
    """fp32 instantiation (README.md:71): exact on integer data while row sums < 2^24, and within
    1e-5 * sum|a x| of the fp32 oracle on real data."""
    for mat in zoo.small_zoo():
        for sigma in (4, 16, 20):
            val, x = M.fill_values(mat.nnz, mat.n, np.float32, seed=3, mode="int")
            fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
            arrays, col_t, val_t, ys = _run(mat, val, x, sigma, mode, dtype=np.float32)
            _check_format(arrays, col_t, val_t, fmt)
            exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
            assert np.array_equal(ys[0], exp), (mat.name, sigma, mode)
        val, x = M.fill_values(mat.nnz, mat.n, np.float32, seed=4, mode="real")
        fmt = oracle.convert(64, 16, mat.m, mat.row_ptr, mat.col, val)
        _, _, _, ys = _run(mat, val, x, 16, mode, dtype=np.float32)
        exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
        scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val), np.abs(x))
        assert np.all(np.abs(ys[0] - exp) <= 1e-5 * np.maximum(scale, 1.0)), mat.name


@pytest.mark.parametrize("ldsy", [0, 2])
@pytest.mark.parametrize("xwin", [0, 2])
def test_lds_x_window_variant(oracle, xwin, ldsy):
    """The LDS x-window variant of the fused kernel (forced on / forced off) must not change a bit:
    same products, same summation order -- on matrices with and without column locality, fp64 and
    fp32, one- and two-packet descriptors."""
    for mat in zoo.small_zoo():
        for sigma, dtype in ((4, np.float64), (16, np.float64), (24, np.float64), (12, np.float32)):
            # integer data: exact, so bit-identical to the oracle whatever the window does
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=30, mode="int")
            fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
            arrays, col_t, val_t, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, xwin=xwin, ldsy=ldsy)
            _check_format(arrays, col_t, val_t, fmt)
            assert np.array_equal(ys[0], _expected_y(oracle, fmt, mat, x, Y_POISON)), (mat.name, sigma, xwin)
            # real data: within tolerance of the oracle (the spill is summed in a different order)
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=31, mode="real")
            fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
            _, _, _, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, xwin=xwin, ldsy=ldsy)
            exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
            scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val), np.abs(x))
            tol = 1e-12 if dtype == np.float64 else 1e-5
            assert np.all(np.abs(ys[0] - exp) <= tol * np.maximum(scale, 1.0)), (mat.name, sigma, xwin)


def test_non_temporal_stream_variant_is_bit_identical():
    """CSR5HIP_OPT_STREAM_NT only changes the cache policy of the column/value loads: forced on and forced off
    must give the same bits on real data (same arithmetic), fp64 and fp32, with and without LDS y segments."""
    for mat in zoo.small_zoo():
        for sigma, dtype in ((4, np.float64), (16, np.float64), (24, np.float64), (12, np.float32)):
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=52, mode="real")
            for ldsy in (0, 2):
                _, _, _, plain = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, xwin=0, ldsy=ldsy, nt=0)
                _, _, _, hinted = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, xwin=0, ldsy=ldsy, nt=2)
                assert np.array_equal(plain[0], hinted[0]), (mat.name, sigma, ldsy)


@pytest.mark.parametrize("ldsy", [0, 2])
def test_two_pass_lds_y_on_off(oracle, ldsy):
    for mat in zoo.small_zoo():
        for sigma in (4, 16, 32):
            val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=41, mode="real")
            fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
            _, _, _, ys = _run(mat, val, x, sigma, H.SPMV_TWO_PASS, ldsy=ldsy)
            exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
            scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val), np.abs(x))
            assert np.all(np.abs(ys[0] - exp) <= 1e-12 * np.maximum(scale, 1.0)), (mat.name, sigma, ldsy)


def test_x_window_auto_selection():
    """Wide band (a gather spreads over ~25 lines of x) -> windows on; uniformly random columns -> off (no coverage);
    columns within +-64 of the diagonal -> off as well in fp64 (covered, but a gather touches <= 8 lines: the window
    would cost more than it saves) (csr5hip_info.x_window_*)."""
    banded = M.nd24k_like(scale=0.02, dtype=np.float64)
    rnd = zoo.small_zoo()[5]  # half-empty, uniform columns over 5000
    rng = np.random.default_rng(4)
    tight = M.csr_from_row_lengths(rng.poisson(8, size=20000).astype(np.int64), 20000, rng, band=1.0, name="tight-band")
    for mat, expect in ((banded, 1), (rnd, 0), (tight, 0)):
        val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=2, mode="int")
        rp, ci, va = _device_csr(mat, val, np.float64)
        xd = torch.from_numpy(x).to(DEV)
        A = H.anonymouslibHandle(mat.m, mat.n)
        A.inputCSR(mat.nnz, rp, ci, va)
        A.setX(xd)
        A.setSigma(16)
        A.setSpmvMode(H.SPMV_FUSED)
        assert A.asCSR5() == 0
        info = A.info()
        assert info.x_window_active == expect, (mat.name, info.x_window_tiles, info.p)
        yd = torch.zeros(mat.m, dtype=torch.float64, device=DEV)
        assert A.spmv(1.0, yd) == 0
        torch.cuda.synchronize()
        assert np.array_equal(yd.cpu().numpy(), oracle_csr(mat, val, x))
        A.destroy()
        A.close()


def oracle_csr(mat, val, x):
    from oracle.csr5_oracle import Oracle
    return Oracle().csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)


def test_runtime_sigma_kernel_and_small_sigma(oracle):
    """sigma = 1..3 run on the run-time-sigma kernel (the reference's switch has no such cases)."""
    for mat in zoo.small_zoo()[:8]:
        val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=6, mode="int")
        for sigma in (1, 2, 3):
            fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
            arrays, col_t, val_t, ys = _run(mat, val, x, sigma, H.SPMV_TWO_PASS)
            _check_format(arrays, col_t, val_t, fmt)
            assert np.array_equal(ys[0], _expected_y(oracle, fmt, mat, x, Y_POISON)), (mat.name, sigma)


def test_empty_matrix_and_state_machine():
    mat = zoo.empty_matrix()
    rp, ci, va = _device_csr(mat, mat.val, np.float64)
    xd = torch.ones(mat.n, dtype=torch.float64, device=DEV)
    yd = torch.full((mat.m,), 3.0, dtype=torch.float64, device=DEV)
    A = H.anonymouslibHandle(mat.m, mat.n)
    assert A.asCSR5() == -1  # _format not set before inputCSR
    assert A.inputCSR(0, rp, ci, va) == 0
    assert A.setX(xd) == 0
    assert A.setSigma(H.ANONYMOUSLIB_AUTO_TUNED_SIGMA) == 0
    assert A.asCSR() == 0  # no-op on CSR
    assert A.asCSR5() == 0
    assert A.asCSR5() == 0  # no-op on CSR5
    assert A.spmv(1.0, yd) == 0
    torch.cuda.synchronize()
    assert torch.all(yd == 3.0)  # nothing to write
    assert A.destroy() == 0
    assert A.setSigma(0) == -101 and A.setSigma(33) == -101
    A.close()


def test_auto_sigma_and_repeat_graph(oracle):
    mat = M.scircuit_like(scale=0.1)
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=8, mode="int")
    rp, ci, va = _device_csr(mat, val, np.float64)
    xd = torch.from_numpy(x).to(DEV)
    yd = torch.zeros(mat.m, dtype=torch.float64, device=DEV)
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    for mode in (H.SPMV_TWO_PASS, H.SPMV_FUSED):
        A = H.anonymouslibHandle(mat.m, mat.n)
        A.inputCSR(mat.nnz, rp, ci, va)
        A.setX(xd)
        assert A.setSigma(H.ANONYMOUSLIB_AUTO_TUNED_SIGMA) == 0
        A.setSpmvMode(mode)
        assert A.asCSR5() == 0
        assert 4 <= A.info().sigma <= 32
        yd.fill_(-1.0)
        assert A.spmv_repeat(1.0, yd, 25) == 0  # 25 captured launches replayed from one hipGraph
        assert A.spmv_repeat(1.0, yd, 25) == 0
        torch.cuda.synchronize()
        assert np.array_equal(yd.cpu().numpy(), ref)
        A.destroy()
        A.close()


def test_full_size_scircuit_properties(oracle):
    """BASELINE config sizes: size-independent properties -- equality with the scalar CSR loop on
    integer data (exact), linearity in x, and asCSR5/asCSR round trip."""
    mat = M.scircuit_like()
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=10, mode="int")
    _, x2 = M.fill_values(mat.nnz, mat.n, np.float64, seed=11, mode="int")
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    ref2 = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x2)
    for mode in (H.SPMV_TWO_PASS, H.SPMV_FUSED):
        for sigma in (H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, 16):
            _, _, _, ys = _run(mat, val, x, sigma, mode, y0=0.0)
            assert np.array_equal(ys[0], ref)
            _, _, _, ys12 = _run(mat, val, x + x2, sigma, mode, y0=0.0)
            assert np.array_equal(ys12[0], ref + ref2)  # linearity, exact on integer data


def test_full_size_webbase_properties(oracle):
    mat = M.webbase_like()
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=12, mode="int")
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    for mode in (H.SPMV_TWO_PASS, H.SPMV_FUSED):
        _, _, _, ys = _run(mat, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, mode, y0=0.0)
        assert np.array_equal(ys[0], ref)


def test_full_size_nd24k_fp32_properties(oracle):
    """BASELINE config 4 (fp32, 28.7 M non-zeros, 399 per row): integer data keeps every fp32 partial sum below
    2^24, so the result must equal the fp32 scalar CSR loop exactly, with and without the LDS x-window."""
    mat = M.nd24k_like()
    val, x = M.fill_values(mat.nnz, mat.n, np.float32, seed=13, mode="int")
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    assert ref.dtype == np.float32 and float(ref.max()) < 2 ** 24
    for mode in (H.SPMV_TWO_PASS, H.SPMV_FUSED):
        for xwin in ((0, 1) if mode == H.SPMV_FUSED else (None,)):
            _, _, _, ys = _run(mat, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, mode, dtype=np.float32, y0=0.0, xwin=xwin)
            assert np.array_equal(ys[0], ref), (mode, xwin)


def test_cli_drop_in(tmp_path):
    """`./spmv file.mtx` (csrc/main.cpp on top of include/anonymouslib_hip.h): the reference CLI's stdout
    lines in the reference's order (CSR5_cuda/main.cu:30,119-384) and its self-check."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "benchmark_spmv_using_csr5_amd", "csrc", "spmv")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    mat = M.example_matrix()
    mat.val[:] = 1.0
    path = tmp_path / "example.mtx"
    M.write_mtx(str(path), mat)
    sym = tmp_path / "sym.mtx"
    sym.write_text("%%MatrixMarket matrix coordinate pattern symmetric\n4 4 5\n1 1\n2 1\n3 2\n4 4\n4 1\n")
    for p, nnz in ((path, mat.nnz), (sym, 8)):
        env = dict(os.environ, CSR5_SEED="7", CSR5_RESULTS=str(tmp_path / "results.csv"))
        out = subprocess.run([exe, str(p)], capture_output=True, text=True, env=env, timeout=300)
        assert out.returncode == 0, out.stderr
        text = out.stdout
        order = ["PRECISION = 64-bit Double Precision", f"--------------{p}--------------",
                 f" ) nnz = {nnz}", "cpu sequential time = ", "Device [0] ", "omega = 64, sigma = ",
                 "CSR->CSR5 malloc time = ", "CSR->CSR5 tile_ptr time = ", "CSR->CSR5 tile_desc time = ",
                 "CSR->CSR5 transpose time = ", "CSR->CSR5 time = ", "CSR5-based SpMV time = ",
                 "Check... PASS!"]
        pos = -1
        for token in order:
            nxt = text.find(token, pos + 1)
            assert nxt > pos, (token, text)
            pos = nxt
    rows = (tmp_path / "results.csv").read_text().strip().splitlines()
    assert len(rows) == 2 and rows[0].startswith(str(path) + ",") and len(rows[0].split(",")) == 9
    assert subprocess.run([exe, str(tmp_path / "missing.mtx")], capture_output=True).returncode == 255  # -1
    bad = tmp_path / "bad.mtx"
    bad.write_text("not a banner\n")
    assert subprocess.run([exe, str(bad)], capture_output=True).returncode == 254  # -2
    cplx = tmp_path / "c.mtx"
    cplx.write_text("%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1.0 0.0\n")
    assert subprocess.run([exe, str(cplx)], capture_output=True).returncode == 253  # -3


@pytest.mark.parametrize("scale", [22, 24])
def test_large_rmat_size_independent_properties(scale):
    """BASELINE-size check (R-MAT scale 22 = 67 M and scale 24 = 268 M non-zeros, the BASELINE.json config, generated
    on the device): on integer data every mode must agree exactly with each other and with an independent
    device-side CSR product (torch.sparse, used as a checker only); asCSR5/asCSR must round-trip the caller's arrays."""
    mat = M.rmat_device(scale, 16, seed=5, rank=0, world=1, device=DEV)
    g = torch.Generator(device=DEV).manual_seed(9)
    val = torch.randint(0, 10, (mat.nnz,), generator=g, device=DEV).to(torch.float64)
    x = torch.randint(0, 10, (mat.n,), generator=g, device=DEV).to(torch.float64)
    ref = torch.sparse_csr_tensor(mat.row_ptr.to(torch.int64), mat.col.to(torch.int64), val,
                                  size=(mat.m, mat.n)) @ x
    col0, val0 = mat.col.clone(), val.clone()
    nonempty = (mat.row_ptr[1:] > mat.row_ptr[:-1])
    for mode in (H.SPMV_TWO_PASS, H.SPMV_FUSED):
        for sigma in ((H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, 7) if scale == 22 else (H.ANONYMOUSLIB_AUTO_TUNED_SIGMA,)):
            A = H.anonymouslibHandle(mat.m, mat.n)
            assert A.inputCSR(mat.nnz, mat.row_ptr, mat.col, val) == 0
            assert A.setX(x) == 0 and A.setSigma(sigma) == 0 and A.setSpmvMode(mode) == 0
            assert A.asCSR5() == 0
            y = torch.full((mat.m,), -3.0, dtype=torch.float64, device=DEV)
            assert A.spmv(1.0, y) == 0
            assert A.spmv(1.0, y) == 0  # second call on the same y: no zeroing needed
            torch.cuda.synchronize()
            assert torch.equal(y[nonempty], ref[nonempty]), (mode, sigma)
            assert A.destroy() == 0
            torch.cuda.synchronize()
            assert torch.equal(mat.col, col0) and torch.equal(val, val0)
            A.close()
    if scale == 22:
        # the plain path at size (no slab structure): 65 536 tiles x sigma 16 -> the auto rule defers the carries
        A = H.anonymouslibHandle(mat.m, mat.n)
        assert A.inputCSR(mat.nnz, mat.row_ptr, mat.col, val) == 0
        assert A.setX(x) == 0 and A.setSigma(H.ANONYMOUSLIB_AUTO_TUNED_SIGMA) == 0 and A.setColumnSlabs(0) == 0
        assert A.asCSR5() == 0
        assert A.info().column_slabs == 0 and A.info().carries_deferred == 1, A.info()
        y = torch.full((mat.m,), -3.0, dtype=torch.float64, device=DEV)
        assert A.spmv(1.0, y) == 0 and A.spmv(1.0, y) == 0
        torch.cuda.synchronize()
        assert torch.equal(y[nonempty], ref[nonempty])
        assert A.destroy() == 0
        A.close()
    # real-valued data at full size, default options (column slabs + hot table where the auto rule picks them):
    # |y - y_ref| <= 1e-12 * sum|a x| against the independent device product, and bit-reproducible run to run
    valr = torch.rand(mat.nnz, generator=g, device=DEV, dtype=torch.float64) * 2 - 1
    xr = torch.rand(mat.n, generator=g, device=DEV, dtype=torch.float64) * 2 - 1
    crow, ccol = mat.row_ptr.to(torch.int64), mat.col.to(torch.int64)
    refr = torch.sparse_csr_tensor(crow, ccol, valr, size=(mat.m, mat.n)) @ xr
    scale_r = torch.sparse_csr_tensor(crow, ccol, valr.abs(), size=(mat.m, mat.n)) @ xr.abs()
    del crow, ccol
    A = H.anonymouslibHandle(mat.m, mat.n)
    assert A.inputCSR(mat.nnz, mat.row_ptr, mat.col, valr) == 0 and A.setX(xr) == 0
    assert A.setSigma(H.ANONYMOUSLIB_AUTO_TUNED_SIGMA) == 0 and A.asCSR5() == 0
    assert A.info().column_slabs >= 8, "the auto rule turns the slab structure on for R-MAT"
    y1 = torch.zeros(mat.m, dtype=torch.float64, device=DEV)
    y2 = torch.zeros(mat.m, dtype=torch.float64, device=DEV)
    assert A.spmv(1.0, y1) == 0 and A.spmv(1.0, y2) == 0
    torch.cuda.synchronize()
    assert torch.equal(y1, y2)
    assert bool(((y1 - refr).abs() <= 1e-12 * torch.clamp(scale_r, min=1.0))[nonempty].all())
    assert A.destroy() == 0
    A.close()


def test_autotune_sigma(oracle):
    """Measured sigma selection: returns a supported sigma, leaves the handle in
    CSR5, and the result is still exact."""
    for mat in (M.scircuit_like(scale=0.2), M.nd24k_like(scale=0.02, dtype=np.float64), zoo.small_zoo()[2]):
        val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=17, mode="int")
        rp, ci, va = _device_csr(mat, val, np.float64)
        xd = torch.from_numpy(x).to(DEV)
        yd = torch.zeros(mat.m, dtype=torch.float64, device=DEV)
        A = H.anonymouslibHandle(mat.m, mat.n)
        A.inputCSR(mat.nnz, rp, ci, va)
        A.setX(xd)
        A.setSpmvMode(H.SPMV_FUSED)
        err, sigma, us = A.autotuneSigma(yd)
        assert err == 0 and sigma in (4, 5, 6, 8, 10, 12, 16, 20, 24, 32) and us > 0
        info = A.info()
        assert info.format == H.ANONYMOUSLIB_FORMAT_CSR5 and info.sigma == sigma
        yd.fill_(5.0)
        assert A.spmv(1.0, yd) == 0
        torch.cuda.synchronize()
        ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
        nonempty = np.diff(mat.row_ptr) > 0
        assert np.array_equal(yd.cpu().numpy()[nonempty], ref[nonempty])
        assert A.destroy() == 0
        torch.cuda.synchronize()
        assert np.array_equal(ci.cpu().numpy(), mat.col)
        A.close()


def test_seeded_fuzz_against_oracle(oracle):
    """Seeded fuzz: random shapes, row-length laws (incl. bursts of empty rows and hub rows), sigma, SpMV
    mode, LDS options, column slabs / hot table and dtype; integer data, so format and y must be bit-identical to the
    oracle."""
    # CSR5_FUZZ_CASES / CSR5_FUZZ_SEED: longer one-off campaigns (scripts/experiments/fuzz_long.sh); defaults = the CI run
    import os
    rng = np.random.default_rng(int(os.environ.get("CSR5_FUZZ_SEED", "20260928")))
    scale = int(os.environ.get("CSR5_FUZZ_SCALE", "1"))
    for case in range(int(os.environ.get("CSR5_FUZZ_CASES", "60"))):
        m = int(rng.integers(1, 4000 * scale))
        n = int(rng.integers(1, 6000 * scale))
        law = case % 5
        if law == 0:
            lens = rng.integers(0, 12, size=m)
        elif law == 1:
            lens = np.floor(rng.pareto(1.3, size=m) * 2).astype(np.int64)
        elif law == 2:
            lens = rng.integers(0, 3, size=m) * (rng.random(m) < 0.3)
            lens[rng.integers(0, m)] = int(rng.integers(500, 20000 * scale))
        elif law == 3:
            lens = np.where(rng.random(m) < 0.5, 0, rng.integers(1, 200, size=m))
        else:
            lens = np.full(m, int(rng.integers(1, 130)))
        lens = np.minimum(lens, 30000 * scale)
        if lens.sum() == 0:
            lens[0] = 1
        band = float(rng.choice([0.0, 0.5, 1.0]))
        mat = M.csr_from_row_lengths(lens, n, rng, band=band, name=f"fuzz{case}")
        dtype = np.float64 if case % 3 else np.float32
        val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=1000 + case, mode="int")
        if dtype == np.float32:  # keep every partial sum below 2^24 so fp32 stays exact
            val = (val % 3).astype(np.float32)
            x = (x % 3).astype(np.float32)
        sigma = int(rng.choice([1, 2, 4, 5, 8, 11, 16, 17, 24, 32]))
        mode = int(rng.integers(0, 2))
        xwin = int(rng.choice([0, 2])) if mode == H.SPMV_FUSED else None
        ldsy = int(rng.choice([0, 2]))
        nt = int(rng.choice([0, 2])) if mode == H.SPMV_FUSED and not xwin else None
        # column slabs (forced; 0 = off) in two cases of five, the LDS hot table (forced) on half of those that can carry it
        slabs = int(rng.choice([2, 4, 8, 16, 32, 64])) if case % 5 in (1, 3) else 0
        hot = int(rng.choice([0, 2])) if slabs % 8 == 0 and slabs and mode == H.SPMV_FUSED else 0
        snap = (case // 5) % 2 if hot else None  # permuted copy of x per spmv (default) or per setX
        narrow = (case // 10) % 2 if hot else None  # fp64: the (integer) values streamed as fp32; fp32 handles ignore it
        # (two draws that selected the round-5 walking kernel, now out of the product: kept so the seeded case stream is unchanged)
        if mode == H.SPMV_FUSED and int(rng.choice([0, 2])):
            rng.choice([0, 1, 2, 3, 7, 40])
        # cut rows finished by the second launch instead of the arrival protocol (forced) on every other fused case
        defer = (0, 2)[(case // 2) % 2] if mode == H.SPMV_FUSED else None
        # sigma 4 .. 8, plain fused kernel: the flagged column words (auto = off at this size) forced on two cases of three
        flagged = 2 if case % 3 != 1 else None
        fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
        arrays, col_t, val_t, ys = _run(mat, val, x, sigma, mode, dtype=dtype, xwin=xwin, ldsy=ldsy, nt=nt, repeat=2,
                                        slabs=slabs, hot=hot, x_snapshot=snap, narrow=narrow, defer=defer, flagged=flagged)
        _check_format(arrays, col_t, val_t, fmt)
        exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
        for y in ys:
            assert np.array_equal(y, exp), (case, m, n, sigma, mode, xwin, ldsy, nt, slabs, hot, dtype, defer,
                                            np.flatnonzero(y != exp)[:5])


def test_multi_tile_rows_stress_cross_xcd_protocol(oracle):
    """Every row spans many tiles (the > 2-party arrival protocol, all 8 XCDs busy): on those rows the
    fused result must be bit-identical to the two-pass result (same partials, same summation order by
    construction) on REAL data, launch after launch -- a stale or torn cross-workgroup read would show
    up here."""
    rng = np.random.default_rng(77)
    lens = rng.integers(1500, 9000, size=260)
    lens[::7] = rng.integers(1, 40, size=lens[::7].size)  # some short rows in between
    mat = M.csr_from_row_lengths(lens, 50_000, rng, band=0.0, name="long-rows")
    for dtype, sigma in ((np.float64, 4), (np.float64, 16), (np.float32, 8)):
        val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=3, mode="real")
        _, _, _, y2 = _run(mat, val, x, sigma, H.SPMV_TWO_PASS, dtype=dtype)
        _, _, _, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, repeat=25)
        long_rows = np.diff(mat.row_ptr) > 64 * sigma * 2  # certainly resolved by the arrival protocol
        for k, y in enumerate(ys):
            assert np.array_equal(y[long_rows], y2[0][long_rows]), (dtype, sigma, k)
            assert np.array_equal(y, ys[0]), (dtype, sigma, k)
        fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
        exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
        scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val), np.abs(x))
        tol = 1e-12 if dtype == np.float64 else 2e-5
        assert np.all(np.abs(y2[0] - exp) <= tol * np.maximum(scale, 1.0))


# ---------------------------------------------------------------------------------------------------
# coupled iterations
# ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("world", [1, 3])
def test_coupled_step_writes_next_x_in_place(oracle, world):
    """One rank's view of a `world`-way coupled step on the GPU: the CSR5 SpMV of the remapped block must put
    exactly A[block] x into the rank's slot of the next x (the collective itself is covered by the gloo tests)."""
    import torch
    from benchmark_spmv_using_csr5_amd import sharding as S
    dev = torch.device("cuda:0")
    mat = M.rmat(scale=12, edge_factor=8, seed=5)
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=6, mode="int")
    y_ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    got = np.zeros(mat.m)
    for rank in range(world):
        cp = S.CoupledSpmv(mat.row_ptr, mat.col, val, mat.n, rank, world)
        run = S.hip_coupled_spmv(dev)
        a = torch.from_numpy(cp.layout.to_padded(x)).to(dev)
        b = torch.full_like(a, -7.0)
        cp.step(run, a, b)           # no process group: the all-gather is skipped, only this rank's slot is written
        torch.cuda.synchronize()
        bh = b.cpu().numpy()
        lo, hi = cp.block.row_lo, cp.block.row_hi
        slot = bh[rank * cp.layout.width: rank * cp.layout.width + (hi - lo)]
        nonempty = np.diff(cp.block.row_ptr) > 0
        assert np.array_equal(slot[nonempty], y_ref[lo:hi][nonempty])
        # nothing outside the slot was touched
        mask = np.ones(bh.size, dtype=bool)
        mask[rank * cp.layout.width: rank * cp.layout.width + (hi - lo)] = False
        assert np.all(bh[mask] == -7.0)
        got[lo:hi] = np.where(nonempty, slot, 0.0)
        run.state["A"].destroy()
        run.state["A"].close()
    assert np.array_equal(got, np.where(np.diff(mat.row_ptr) > 0, y_ref, 0.0))


@pytest.mark.gpu
def test_coupled_power_iteration_single_gpu(oracle):
    import torch
    from benchmark_spmv_using_csr5_amd import sharding as S
    dev = torch.device("cuda:0")
    mat = M.rmat(scale=12, edge_factor=8, seed=5)
    val, x0 = M.fill_values(mat.nnz, mat.n, np.float64, seed=6, mode="pos")
    x0 = x0 / np.linalg.norm(x0)
    cp = S.CoupledSpmv(mat.row_ptr, mat.col, val, mat.n, 0, 1)
    run = S.hip_coupled_spmv(dev)
    a = torch.from_numpy(cp.layout.to_padded(x0)).to(dev)
    b = torch.zeros_like(a)
    xk, lam = cp.power_iteration(run, a, b, iters=15)
    torch.cuda.synchronize()
    x, lam_ref = x0.copy(), None
    for _ in range(15):
        y = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
        lam_ref = float(np.dot(x, y) / np.dot(x, x))
        x = y / np.linalg.norm(y)
    got = cp.layout.from_padded(xk.cpu().numpy())
    assert np.max(np.abs(got - x)) < 1e-10          # fp64, different summation order per row
    assert abs(float(lam) - lam_ref) < 1e-9 * abs(lam_ref)
    run.state["A"].destroy()
    run.state["A"].close()


@pytest.mark.gpu
def test_coupled_iteration_from_one_hipgraph(oracle):
    """A coupled iteration (SpMV -> [all-gather] -> dot, norm, scale) captured in ONE hipGraph and replayed: same
    result as the eager loop, bit for bit."""
    import torch
    from benchmark_spmv_using_csr5_amd import sharding as S
    dev = torch.device("cuda:0")
    mat = M.rmat(scale=12, edge_factor=8, seed=5)
    val, x0 = M.fill_values(mat.nnz, mat.n, np.float64, seed=6, mode="pos")
    x0 = x0 / np.linalg.norm(x0)
    cp = S.CoupledSpmv(mat.row_ptr, mat.col, val, mat.n, 0, 1)
    eager = S.hip_coupled_spmv(dev)
    a = torch.from_numpy(cp.layout.to_padded(x0)).to(dev)
    xe, lam_e = cp.power_iteration(eager, a, torch.zeros_like(a), iters=10)
    torch.cuda.synchronize()
    graphed = S.hip_coupled_spmv(dev)
    a2 = torch.from_numpy(cp.layout.to_padded(x0)).to(dev)
    xg, lam_g = cp.power_iteration_graph(graphed, a2, torch.zeros_like(a2), iters=10)
    torch.cuda.synchronize()
    assert torch.equal(xe, xg) and float(lam_e) == float(lam_g)
    for run in (eager, graphed):
        run.state["A"].destroy()
        run.state["A"].close()


@pytest.mark.gpu
def test_coupled_steps_without_normalisation_define_empty_rows(oracle):
    """x_{k+1} = A x_k EXACTLY, three raw steps on a matrix whose rows are half empty, starting from ping-pong
    buffers full of non-zero garbage: the slot of an empty row must become 0, not keep what it held two steps ago
    (the library zeroes empty rows for the coupled path, CSR5HIP_OPT_ZERO_EMPTY_ROWS)."""
    import torch
    from benchmark_spmv_using_csr5_amd import sharding as S
    dev = torch.device("cuda:0")
    mat = M.rmat(scale=11, edge_factor=4, seed=9)
    assert (np.diff(mat.row_ptr) == 0).mean() > 0.3
    val, x0 = M.fill_values(mat.nnz, mat.n, np.float64, seed=6, mode="int")
    val = (val % 3).astype(np.float64)
    cp = S.CoupledSpmv(mat.row_ptr, mat.col, val, mat.n, 0, 1)
    run = S.hip_coupled_spmv(dev)
    a = torch.from_numpy(cp.layout.to_padded(x0)).to(dev)
    b = torch.full_like(a, 12345.0)  # stale contents of the other buffer
    xk, _ = cp.power_iteration(run, a, b, iters=3, normalise=False)
    torch.cuda.synchronize()
    x = x0.copy()
    for _ in range(3):
        x = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    assert np.array_equal(cp.layout.from_padded(xk.cpu().numpy()), x)
    run.state["A"].destroy()
    run.state["A"].close()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
@pytest.mark.parametrize("sigma", [4, 16])
def test_fused_long_run_path_deterministic(oracle, sigma, dtype):
    """Rows spanning MORE than 64 tiles take the parked-partial path of the fused kernel (carry_meta bit 26 +
    k_calibrate<LONG_ONLY>): one row of 70 * 64 * sigma non-zeros between short rows, and a 10 M-nnz hub row.
    Fused and two-pass must agree bit for bit on real data (same summation order by construction), both within the
    tolerance of the oracle, and exactly on integer data."""
    T = 64 * sigma
    cases = [[3, 0, 70 * T + 17, 5, 1, 0, 2], [1] * 50 + [66 * T] + [2] * 30 + [65 * T + 1, 0, 7]]
    if sigma == 16 and dtype == np.float64:
        cases.append([2, 10_000_000, 1, 0, 4])
    for k, lens in enumerate(cases):
        rng = np.random.default_rng(100 + k)
        mat = M.csr_from_row_lengths(np.asarray(lens), 50000, rng, band=0.0, name=f"longrun{k}")
        for fill in ("int", "real"):
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=3, mode=fill)
            if dtype == np.float32 and fill == "int":
                val, x = (val % 2).astype(np.float32), (x % 2).astype(np.float32)
            if mat.nnz > 5_000_000:
                exp = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x).astype(np.float64)
            else:
                fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
                exp = _expected_y(oracle, fmt, mat, x, 0.0).astype(np.float64)
            scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val).astype(np.float64), np.abs(x).astype(np.float64))
            _, _, _, yf = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, y0=0.0, repeat=2, slabs=0)
            _, _, _, yt = _run(mat, val, x, sigma, H.SPMV_TWO_PASS, dtype=dtype, y0=0.0, slabs=0)
            assert np.array_equal(yf[0], yf[1]) and np.array_equal(yf[0], yt[0]), (k, fill, "fused == two-pass, bit for bit")
            tol = (1e-12 if dtype == np.float64 else 2e-5) * np.maximum(scale, 1.0)
            nonempty = np.diff(mat.row_ptr) > 0
            if fill == "int" and (dtype == np.float64 or mat.nnz < 2 ** 23):
                assert np.array_equal(yf[0].astype(np.float64)[nonempty], exp[nonempty]), (k, fill)
            else:
                assert np.all(np.abs(yf[0].astype(np.float64) - exp)[nonempty] <= tol[nonempty]), (k, fill)


@pytest.mark.gpu
@pytest.mark.parametrize("workload", ["scircuit", "webbase", "nd24k"])
def test_full_size_real_valued_within_tolerance(oracle, workload):
    """One real-valued (uniform(-1,1)) FULL-SIZE run per single-GPU BASELINE config against the oracle's CSR5 SpMV:
    fp64 within 1e-6 relative where the row is not ill-conditioned and 1e-12 * sum|a x| everywhere; fp32 (nd24k):
    1e-5 * sum|a x|.  Default options (so the column-slab path where the auto rule selects it)."""
    dtype = np.float32 if workload == "nd24k" else np.float64
    mat = {"scircuit": M.scircuit_like, "webbase": M.webbase_like, "nd24k": M.nd24k_like}[workload]()
    val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=21, mode="real")
    sigma = _capi.load().csr5hip_auto_sigma(mat.m, mat.nnz, 0)
    fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
    exp = _expected_y(oracle, fmt, mat, x, 0.0).astype(np.float64)
    scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val).astype(np.float64), np.abs(x).astype(np.float64))
    _, _, _, ys = _run(mat, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, H.SPMV_FUSED, dtype=dtype, y0=0.0, repeat=2)
    y = ys[0].astype(np.float64)
    assert np.array_equal(ys[0], ys[1])
    tol = 1e-12 if dtype == np.float64 else 1e-5
    assert np.all(np.abs(y - exp) <= tol * np.maximum(scale, 1.0))
    if dtype == np.float64:
        well = np.abs(exp) >= 1e-3 * scale  # rows without heavy cancellation: the 1e-6 relative bar of north_star
        assert np.all(np.abs(y - exp)[well] <= 1e-6 * np.abs(exp)[well])


# ---------------------------------------------------------------------------------------------------
# checkpoint of the converted matrix
# ---------------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_save_load_roundtrip(tmp_path, dtype):
    import torch
    from benchmark_spmv_using_csr5_amd.handle import anonymouslibHandle
    dev = torch.device("cuda:0")
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    mat = M.webbase_like(scale=0.02, seed=8)       # empty rows -> offset arrays are exercised
    val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=2, mode="int")
    rp, ci, va, xd = (torch.from_numpy(a).to(dev) for a in (mat.row_ptr, mat.col, val, x))
    y0 = torch.zeros(mat.m, dtype=tdt, device=dev)
    A = anonymouslibHandle(mat.m, mat.n, dtype=np.dtype(dtype).name)
    assert A.inputCSR(mat.nnz, rp, ci, va) == 0 and A.setX(xd) == 0
    path = str(tmp_path / "a.csr5")
    assert A.save(path) != 0                        # still CSR: nothing to checkpoint
    assert A.setSigma(6) == 0 and A.asCSR5() == 0
    assert A.spmv(1.0, y0) == 0
    torch.cuda.synchronize()
    before = A.csr5_arrays()
    assert A.save(path) == 0

    B = anonymouslibHandle.load(path)
    ib = B.info()
    assert (ib.format, ib.m, ib.n, ib.nnz, ib.sigma, ib.p) == (_capi.FORMAT_CSR5, mat.m, mat.n, mat.nnz, 6, A.info().p)
    after = B.csr5_arrays()
    for k in ("tile_ptr", "tile_desc", "offset_ptr", "offset"):
        assert np.array_equal(before[k], after[k]), k
    y1 = torch.full_like(y0, -1.0)
    y1[torch.from_numpy(np.diff(mat.row_ptr) == 0).to(dev)] = 0
    assert B.setX(xd) == 0 and B.spmv(1.0, y1) == 0
    torch.cuda.synchronize()
    assert torch.equal(y0, y1)
    # back to CSR on the loaded arrays: the caller's original matrix
    assert B.asCSR() == 0
    host = B.arrays.to_host()
    assert np.array_equal(host.row_ptr, mat.row_ptr) and np.array_equal(host.col, mat.col)
    assert np.array_equal(host.val, val)
    B.close()
    A.destroy()
    A.close()
    # damaged files are rejected before anything is built
    bad = tmp_path / "bad.csr5"
    bad.write_bytes(open(path, "rb").read()[:200])
    with pytest.raises(RuntimeError):
        anonymouslibHandle.load(str(bad))
    bad.write_bytes(b"NOTCSR5!" + bytes(100))
    with pytest.raises(RuntimeError):
        anonymouslibHandle.load(str(bad))


@pytest.mark.gpu
def test_handle_lifecycle_does_not_leak_device_memory(tmp_path):
    """create -> convert -> eager + graph SpMV -> sigma change -> save/load -> destroy -> free, many times:
    the device's free memory must come back (buffers, graphs, events, checkpoint arrays)."""
    import torch
    from benchmark_spmv_using_csr5_amd.handle import anonymouslibHandle
    dev = torch.device("cuda:0")
    mat = M.scircuit_like(scale=0.2)
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=3, mode="int")
    rp, ci, va, xd = (torch.from_numpy(a).to(dev) for a in (mat.row_ptr, mat.col, val, x))
    yd = torch.zeros(mat.m, dtype=torch.float64, device=dev)
    path = str(tmp_path / "c.csr5")

    def cycle():
        A = anonymouslibHandle(mat.m, mat.n)
        assert A.inputCSR(mat.nnz, rp, ci, va) == 0 and A.setX(xd) == 0
        for sigma in (5, 16):
            assert A.setSigma(sigma) == 0 and A.asCSR5() == 0
            assert A.spmv(1.0, yd) == 0 and A.spmv_repeat(1.0, yd, 20) == 0
            assert A.asCSR() == 0
        assert A.asCSR5() == 0 and A.save(path) == 0
        B = anonymouslibHandle.load(path)
        assert B.setX(xd) == 0 and B.spmv(1.0, yd) == 0
        torch.cuda.synchronize()
        B.close()
        assert A.destroy() == 0
        A.close()

    for _ in range(3):
        cycle()
    torch.cuda.synchronize()
    free0, _ = torch.cuda.mem_get_info()
    for _ in range(40):
        cycle()
    torch.cuda.synchronize()
    free1, _ = torch.cuda.mem_get_info()
    assert free0 - free1 < 32 << 20, f"{(free0 - free1) >> 20} MiB of device memory did not come back"
    assert np.array_equal(ci.cpu().numpy(), mat.col)


@pytest.mark.gpu
def test_plain_c_host_program(tmp_path):
    """tests/c/abi_smoke.c: a C99 program on the C ABI alone (what a foreign-language binding would do)."""
    import subprocess
    from tests.test_host import _build_c_host
    out = subprocess.run([_build_c_host(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, (out.stdout, out.stderr)
    assert "mismatches=0" in out.stdout


@pytest.mark.gpu
def test_spmv_is_capturable_in_a_callers_graph(oracle):
    """csr5hip_spmv only enqueues kernels on the handle's stream (no allocation, no synchronisation), so a caller
    can record it into its own hipGraph next to other work -- here a torch CUDA graph: scale x, SpMV, add."""
    mat = M.scircuit_like(scale=0.1)
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=5, mode="int")
    rp, ci, va = _device_csr(mat, val, np.float64)
    xd = torch.from_numpy(x).to(DEV)
    yd = torch.zeros(mat.m, dtype=torch.float64, device=DEV)
    zd = torch.zeros_like(yd)
    side = torch.cuda.Stream(device=DEV)
    for mode in (H.SPMV_FUSED, H.SPMV_TWO_PASS):
        A = H.anonymouslibHandle(mat.m, mat.n, stream=side.cuda_stream)
        assert A.inputCSR(mat.nnz, rp, ci, va) == 0 and A.setX(xd) == 0
        assert A.setSpmvMode(mode) == 0 and A.setSigma(8) == 0 and A.asCSR5() == 0
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            assert A.spmv(1.0, yd) == 0          # warm-up outside the capture
        side.synchronize()
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=side):
            xd.mul_(2.0)
            assert A.spmv(1.0, yd) == 0
            zd.copy_(yd).add_(1.0)
        for rep in range(3):
            xd.copy_(torch.from_numpy(x).to(DEV))
            yd.fill_(-5.0)
            torch.cuda.synchronize()
            graph.replay()
            torch.cuda.synchronize()
            ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, 2.0 * x)
            nonempty = np.diff(mat.row_ptr) > 0
            assert np.array_equal(yd.cpu().numpy()[nonempty], ref[nonempty]), (mode, rep)
            assert np.array_equal(zd.cpu().numpy()[nonempty], ref[nonempty] + 1.0)
        del graph
        A.destroy()
        A.close()


@pytest.mark.gpu
def test_batch_harness_writes_one_row_per_matrix(tmp_path):
    """scripts/bench_batch.py (the reference's avx512 CLI appends `file,GFlops` to results.csv,
    CSR5_avx512/main.cpp:105-110): two Matrix Market files in -> two JSON lines and two CSV rows out, with the matrix's
    dimensions, sigma, GFLOPS and roofline fraction; a second invocation APPENDS."""
    import csv
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    mats = [M.scircuit_like(scale=0.05), M.example_matrix()]
    files = []
    for k, mat in enumerate(mats):
        mat.val[:] = 1.0
        path = tmp_path / f"m{k}.mtx"
        M.write_mtx(str(path), mat)
        files.append(str(path))
    out = str(tmp_path / "results")
    cmd = [sys.executable, os.path.join(root, "scripts", "bench_batch.py"), *files, "--out", out, "--steps", "20", "--warmup", "5",
           "--no-cpu-baseline"]
    run = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert run.returncode == 0, run.stderr
    lines = [json.loads(l) for l in open(out + ".jsonl")]
    rows = list(csv.DictReader(open(out + ".csv")))
    assert len(lines) == 2 and len(rows) == 2
    for mat, f, d, r in zip(mats, files, lines, rows):
        assert d["file"] == f and r["file"] == f
        assert (int(r["m"]), int(r["n"]), int(r["nnz"])) == (mat.m, mat.n, mat.nnz)
        assert d["config"]["nnz_per_gpu"] == mat.nnz and d["n_gpus"] == 1 and r["dtype"] == "f64"
        assert 4 <= int(r["sigma"]) <= 32 and float(r["gflops"]) > 0 and 0 < float(r["roof_frac"]) < 1
        assert "matrix market file" in d["data"]
    run = subprocess.run(cmd[:3] + ["--out", out, "--steps", "20", "--warmup", "5", "--no-cpu-baseline"], capture_output=True,
                         text=True, timeout=900)
    assert run.returncode == 0, run.stderr
    assert len(open(out + ".jsonl").readlines()) == 3 and len(list(csv.DictReader(open(out + ".csv")))) == 3


@pytest.mark.gpu
def test_deferred_carries(oracle):
    """CSR5HIP_OPT_DEFER_CARRIES: no tile finishes its neighbour's short spill, the parties of every cut row park their partials
    with plain stores and a second launch (k_calibrate) adds them in tile order.  Forced on the zoo and on a matrix whose rows
    span many tiles: exact against the oracle on integer data; on real data bit-identical to the two-pass mode (same partials,
    same order) launch after launch, and within the tolerance of the in-launch protocol; every kernel family (x-window, 16-bit
    codes, LDS y, NT).  Auto follows tiles, sigma and the average row length."""
    rng = np.random.default_rng(91)
    lens = rng.integers(1500, 9000, size=200)
    lens[::5] = rng.integers(0, 40, size=lens[::5].size)
    long_rows = M.csr_from_row_lengths(lens, 50_000, rng, band=0.0, name="long-rows")
    mats = zoo.small_zoo() + [long_rows, M.nd24k_like(scale=0.05, dtype=np.float64)]
    for mat in mats:
        for sigma, dtype, kw in ((4, np.float64, {}), (16, np.float64, dict(xwin=2)), (7, np.float64, dict(ldsy=2)),
                                 (24, np.float32, dict(xwin=2)), (16, np.float32, dict(nt=2)), (32, np.float32, {})):
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=71, mode="int")
            if dtype == np.float32:
                val, x = (val % 3).astype(np.float32), (x % 3).astype(np.float32)
            fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
            info = {}
            arrays, col_t, val_t, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, slabs=0, defer=2, repeat=2,
                                            info_out=info, **kw)
            _check_format(arrays, col_t, val_t, fmt)
            assert info["carries_deferred"] == (1 if fmt.p > 1 else 0), info
            exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
            assert np.array_equal(ys[0], exp) and np.array_equal(ys[1], exp), (mat.name, sigma, np.dtype(dtype).name, info)
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=72, mode="real")
            info_off = {}
            _, _, _, yd = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, slabs=0, defer=2, repeat=3, **kw)
            _, _, _, yo = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, slabs=0, defer=0, info_out=info_off, **kw)
            _, _, _, y2 = _run(mat, val, x, sigma, H.SPMV_TWO_PASS, dtype=dtype, slabs=0, defer=0)
            assert info_off["carries_deferred"] == 0
            nonempty = np.diff(mat.row_ptr) > 0
            for y in yd:
                assert np.array_equal(y, yd[0]), (mat.name, sigma, "launch after launch")
            assert np.array_equal(yd[0][nonempty], y2[0][nonempty]), (mat.name, sigma, "deferred == two-pass, bit for bit")
            scale = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, np.abs(val).astype(np.float64), np.abs(x).astype(np.float64))
            tol = (1e-12 if dtype == np.float64 else 1e-5) * np.maximum(scale, 1.0)
            assert np.all(np.abs(yd[0].astype(np.float64) - yo[0].astype(np.float64))[nonempty] <= tol[nonempty]), (mat.name, sigma)
    # auto: long rows from 3 000 tiles on; short rows only when tiles x sigma is large; small matrices never
    for mk, dtype, sigma, expect in ((lambda: M.nd24k_like(scale=0.25, dtype=np.float32), np.float32, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, 1),
                                     (lambda: M.nd24k_like(scale=0.05, dtype=np.float32), np.float32, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, 0),
                                     (lambda: M.scircuit_like(), np.float64, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, 0),
                                     (lambda: M.rmat(18, 16, seed=4), np.float64, 16, 0)):     # 4 096 tiles x 16: short rows,
        # too few tiles (the positive short-row case needs >= 32 M non-zeros: test_large_rmat_size_independent_properties)
        mat = mk()
        val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=73, mode="int")
        if dtype == np.float32:
            val, x = (val % 3).astype(np.float32), (x % 3).astype(np.float32)
        info = {}
        _, _, _, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, slabs=0, info_out=info)
        assert info["carries_deferred"] == expect, (mat.name, info)
        nonempty = np.diff(mat.row_ptr) > 0
        ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val.astype(np.float64), x.astype(np.float64))
        assert np.array_equal(ys[0].astype(np.float64)[nonempty], ref[nonempty]), mat.name


@pytest.mark.gpu
def test_flagged_column_words(oracle):
    """CSR5HIP_OPT_FLAGGED_COLUMNS: at sigma 4 .. 8 the plain fused kernel streams a kernel-side copy of the tile-ordered
    column_index with the element's row-start flag in bit 31 and loads no descriptor word (y_offset recomputed from the flags).
    Same gathers, same arithmetic: bit-identical to the column_index + tile_desc kernel on real data, exact against the oracle
    on integer data; the zoo (empty rows, hub rows, one-row tiles, ragged tails), every kernel family it has (LDS y, NT streams,
    deferred carries); the exposed format arrays -- column_index among them -- stay bit-exact; other sigmas, the x-window kernel
    and the two-pass mode keep the descriptor words."""
    mats = zoo.small_zoo() + [M.scircuit_like(scale=0.2), M.webbase_like(scale=0.05)]
    for mat in mats:
        for sigma, dtype, kw in ((4, np.float64, {}), (6, np.float64, dict(ldsy=2)), (6, np.float64, dict(ldsy=0, defer=2)),
                                 (8, np.float64, dict(nt=2)), (5, np.float32, {}), (8, np.float32, dict(ldsy=2, nt=2)),
                                 (7, np.float32, dict(defer=2))):
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=81, mode="int")
            if dtype == np.float32:
                val, x = (val % 3).astype(np.float32), (x % 3).astype(np.float32)
            fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
            info = {}
            arrays, col_t, val_t, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, slabs=0, xwin=0, repeat=2, info_out=info,
                                            flagged=2, **kw)
            _check_format(arrays, col_t, val_t, fmt)  # (column_index is only read: still the reference's transposed array)
            assert info["flagged_columns"] == (1 if fmt.p > 1 else 0), (mat.name, sigma, info)
            exp = _expected_y(oracle, fmt, mat, x, Y_POISON)
            assert np.array_equal(ys[0], exp) and np.array_equal(ys[1], exp), (mat.name, sigma, np.dtype(dtype).name, kw)
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=82, mode="real")
            _, _, _, yf = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, slabs=0, xwin=0, repeat=2, flagged=2, **kw)
            info = {}
            _, _, _, yd = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, slabs=0, xwin=0, flagged=0, info_out=info, **kw)
            assert info["flagged_columns"] == 0
            assert np.array_equal(yf[0], yd[0]) and np.array_equal(yf[0], yf[1]), (mat.name, sigma, "bit-identical")
    mat = M.scircuit_like(scale=0.2)
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=83, mode="int")
    for sigma, mode, kw, expect in ((12, H.SPMV_FUSED, dict(xwin=0, flagged=2), 0), (6, H.SPMV_TWO_PASS, dict(flagged=2), 0),
                                    (6, H.SPMV_FUSED, dict(xwin=2, flagged=2), 0),
                                    (H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, H.SPMV_FUSED, dict(flagged=2), 1),  # (auto sigma of 5.6 per row: 6)
                                    (6, H.SPMV_FUSED, {}, 0)):  # auto: a 2-MB matrix is latency-bound, the saving is bytes -> off
        info = {}
        _, _, _, ys = _run(mat, val, x, sigma, mode, slabs=0, info_out=info, **kw)
        assert info["flagged_columns"] == expect, (sigma, mode, kw, info)
        assert np.array_equal(ys[0], _expected_y(oracle, oracle.convert(64, info["sigma"], mat.m, mat.row_ptr, mat.col, val), mat, x, Y_POISON))
    # switched on a converted handle, and across asCSR / asCSR5
    rp, ci, va = _device_csr(mat, val, np.float64)
    xd = torch.from_numpy(x).to(DEV)
    yd = torch.full((mat.m,), Y_POISON, dtype=torch.float64, device=DEV)
    ref = oracle.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
    A = H.anonymouslibHandle(mat.m, mat.n)
    assert A.inputCSR(mat.nnz, rp, ci, va) == 0 and A.setX(xd) == 0 and A.setSigma(6) == 0 and A.setFlaggedColumns(0) == 0
    assert A.asCSR5() == 0 and A.info().flagged_columns == 0
    for on in (2, 0, 2):
        assert A.setFlaggedColumns(on) == 0 and A.info().flagged_columns == on // 2
        yd.fill_(Y_POISON)
        assert A.spmv(1.0, yd) == 0 and A.spmv_repeat(1.0, yd, 2) == 0
        torch.cuda.synchronize()
        assert np.array_equal(yd.cpu().numpy(), ref), on
    assert A.asCSR() == 0 and A.setSigma(8) == 0 and A.asCSR5() == 0 and A.info().flagged_columns == 1
    assert A.spmv(1.0, yd) == 0
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy(), ref)
    assert A.setSpmvMode(H.SPMV_TWO_PASS) == 0 and A.info().flagged_columns == 0 and A.spmv(1.0, yd) == 0
    assert A.setSpmvMode(H.SPMV_FUSED) == 0 and A.info().flagged_columns == 1 and A.spmv(1.0, yd) == 0
    torch.cuda.synchronize()
    assert np.array_equal(yd.cpu().numpy(), ref)
    assert A.destroy() == 0
    assert np.array_equal(ci.cpu().numpy(), mat.col), "the caller's column_index comes back untouched"
    A.close()


@pytest.mark.gpu
def test_narrow_column_codes(oracle):
    """CSR5HIP_OPT_NARROW_COLUMNS: the x-window kernel streams 16-bit column codes (column - smallest column of the tile) when
    every tile spans < 32 768 columns (15 bits of column, bit 15 = the element's row-start flag: no descriptor load).  Same gathers, same arithmetic: bit-identical to the 32-bit column stream on real data,
    exact against the oracle on integer data; matrices with and without empty rows / hub rows (the zoo: n < 32 768, so every
    tile is narrow once the window kernel is forced); a matrix with a tile spanning > 32 767 columns keeps the 32-bit stream."""
    mats = zoo.small_zoo() + [M.nd24k_like(scale=0.05, dtype=np.float64)]
    for mat in mats:
        for sigma, dtype in ((8, np.float64), (16, np.float64), (24, np.float64), (16, np.float32), (32, np.float32)):
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=61, mode="int")
            if dtype == np.float32:
                val, x = (val % 3).astype(np.float32), (x % 3).astype(np.float32)
            fmt = oracle.convert(64, sigma, mat.m, mat.row_ptr, mat.col, val)
            info = {}
            arrays, col_t, val_t, ys = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, xwin=2, slabs=0, info_out=info)
            _check_format(arrays, col_t, val_t, fmt)  # the format arrays (incl. the 32-bit column_index) are untouched
            assert info["narrow_columns"] == (1 if fmt.p > 1 else 0), (mat.name, sigma, info)
            assert np.array_equal(ys[0], _expected_y(oracle, fmt, mat, x, Y_POISON)), (mat.name, sigma, np.dtype(dtype).name)
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=62, mode="real")
            _, _, _, y16 = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, xwin=2, slabs=0, repeat=2)
            info = {}
            _, _, _, y32 = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, xwin=2, slabs=0, narrow_cols=0, info_out=info)
            assert info["narrow_columns"] == 0
            assert np.array_equal(y16[0], y32[0]) and np.array_equal(y16[0], y16[1]), (mat.name, sigma, "bit-identical")
            # x one element off a 16-byte boundary: the window is staged element by element instead of in 16-byte pieces
            _, _, _, yu = _run(mat, val, x, sigma, H.SPMV_FUSED, dtype=dtype, xwin=2, slabs=0, x_misaligned=True)
            assert np.array_equal(yu[0], y16[0]), (mat.name, sigma, "element-aligned x")
    # auto: the banded stand-in gets windows AND codes without being asked; sigma 6 (no instantiation) and a wide matrix do not
    nd = M.nd24k_like(scale=0.05, dtype=np.float32)
    val, x = M.fill_values(nd.nnz, nd.n, np.float32, seed=63, mode="int")
    val, x = (val % 3).astype(np.float32), (x % 3).astype(np.float32)
    info = {}
    _, _, _, ys = _run(nd, val, x, H.ANONYMOUSLIB_AUTO_TUNED_SIGMA, H.SPMV_FUSED, dtype=np.float32, info_out=info)
    assert info["sigma"] == 16 and info["x_window_active"] == 1 and info["narrow_columns"] == 1, info  # (fp32, 399 per row: u = 16)
    assert np.array_equal(ys[0].astype(np.float64), oracle.csr_spmv(nd.m, nd.row_ptr, nd.col, val.astype(np.float64), x.astype(np.float64)))
    rng = np.random.default_rng(5)
    wide = M.csr_from_row_lengths(rng.integers(1, 40, size=30000), 300_000, rng, band=0.0, name="wide")
    val, x = M.fill_values(wide.nnz, wide.n, np.float64, seed=64, mode="int")
    info = {}
    _, _, _, ys = _run(wide, val, x, 16, H.SPMV_FUSED, xwin=2, slabs=0, info_out=info)
    assert info["x_window_active"] == 1 and info["narrow_columns"] == 0, info
    assert np.array_equal(ys[0][np.diff(wide.row_ptr) > 0], oracle.csr_spmv(wide.m, wide.row_ptr, wide.col, val, x)[np.diff(wide.row_ptr) > 0])

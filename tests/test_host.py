"""CPU tests (no GPU): host logic and the C-ABI surface.  No compute entry point is called here."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "benchmark_spmv_using_csr5_amd")

from benchmark_spmv_using_csr5_amd import _capi, matrices as M, sharding as S  # noqa: E402


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "csr5hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(csr5hip_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    """libcsr5hip.so loads without a GPU and exports exactly what include/csr5hip.h declares."""
    declared = _header_symbols()
    bound = sorted(name for name, _, _ in _capi.SYMBOLS)
    assert declared == bound, set(declared) ^ set(bound)
    lib = _capi.load()  # raises if the library or any symbol is missing
    for name in declared:
        assert getattr(lib, name) is not None
    assert b"gfx950" in lib.csr5hip_version()


def test_handle_host_logic_without_gpu():
    """State machine and argument checks that never touch the device (anonymouslib_cuda.h:61-76,
    :262-271, :294-318)."""
    lib = _capi.load()
    h = C.c_void_p()
    assert lib.csr5hip_create(C.byref(h), 10, 10, 7) == _capi.UNSUPPORTED_VALUE_TYPE
    assert lib.csr5hip_create(C.byref(h), 10, 10, _capi.F64) == 0
    assert lib.csr5hip_as_csr5(h) == _capi.UNKOWN_FORMAT          # before inputCSR
    assert lib.csr5hip_input_csr(h, 100, None, None, None) == 0
    assert lib.csr5hip_spmv(h, 1.0, C.c_void_p(8)) == _capi.UNSUPPORTED_CSR_SPMV  # format is CSR
    assert lib.csr5hip_as_csr(h) == 0                             # no-op on CSR
    assert lib.csr5hip_set_sigma(h, 0) == _capi.INVALID_ARGUMENT
    assert lib.csr5hip_set_sigma(h, 33) == _capi.INVALID_ARGUMENT
    assert lib.csr5hip_set_sigma(h, 16) == 0
    assert lib.csr5hip_set_sigma(h, _capi.AUTO_TUNED_SIGMA) == 0
    info = _capi.Csr5Info()
    assert lib.csr5hip_get_info(h, C.byref(info)) == 0
    assert (info.m, info.n, info.nnz, info.omega, info.format) == (10, 10, 100, 64, _capi.FORMAT_CSR)
    assert info.sigma == lib.csr5hip_auto_sigma(10, 100, _capi.F64)
    assert lib.csr5hip_set_option(h, 99, 0) == _capi.INVALID_ARGUMENT
    assert lib.csr5hip_destroy(h) == 0
    assert lib.csr5hip_free(h) == 0


def test_auto_sigma_rule_shape():
    """sigma = r if k<=r; k if k<=s; s if k<=t; else u with k = nnz/m (anonymouslib_cuda.h:297-313)."""
    lib = _capi.load()
    sig = [lib.csr5hip_auto_sigma(1000, 1000 * k, _capi.F64) for k in (0, 1, 4, 5, 9, 16, 17, 200, 256, 257, 5000)]
    assert all(4 <= s <= 32 for s in sig)
    assert sig[0] == sig[1] == sig[2] == sig[3] == 6 and sig[4] == 9 and sig[5] == 16 and sig[8] == 16 and sig[9] == 16
    # fp32 has its own table, as the reference keeps one per architecture and precision: r = 8
    s32 = [lib.csr5hip_auto_sigma(1000, 1000 * k, _capi.F32) for k in (0, 1, 4, 5, 8, 9, 16, 17, 200, 256, 257, 5000)]
    assert s32[:5] == [8] * 5 and s32[5] == 9 and s32[6] == 16 and s32[9] == 16 and s32[10] == 16 and s32[11] == 16  # u = 16


def test_product_never_touches_the_oracle():
    """The product path must not import, link or call anything under oracle/ (no CPU fallback)."""
    forbidden = ("import oracle", "from oracle", "csr5_oracle", "csr5oracle", "libref_", "oracle/")
    for base, _, files in os.walk(PKG):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".cpp")) or f == "Makefile":
                text = open(os.path.join(base, f), errors="ignore").read()
                for tok in forbidden:
                    assert tok not in text, (f, tok)
    for hdr in ("csr5hip.h", "anonymouslib_hip.h"):
        assert "oracle" not in open(os.path.join(ROOT, "include", hdr)).read().lower()
    ldd = subprocess.run(["ldd", _capi.LIB_PATH], capture_output=True, text=True).stdout
    assert "csr5oracle" not in ldd and "libref_" not in ldd


def test_algorithmic_bytes_match_baseline_table():
    # scircuit fp64 14.9 MB, webbase 57.3 MB, nd24k fp32 230.6 MB
    assert round(M.algorithmic_bytes(170_998, 170_998, 958_936, 8) / 1e6, 1) == 14.9
    assert round(M.algorithmic_bytes(1_000_005, 1_000_005, 3_105_536, 8) / 1e6, 1) == 57.3
    assert round(M.algorithmic_bytes(72_000, 72_000, 28_715_634, 4) / 1e6, 1) == 230.6
    assert M.reference_getB(100, 1000, 8) == (100 + 1 + 1000) * 4 + (2 * 1000 + 100) * 8


def test_mtx_ingest_follows_reference_cli(tmp_path):
    """General / symmetric / pattern banners, 1-based indices, file order inside rows, mirrored
    off-diagonals right after their original, values discarded (main.cpp:180-275, 283-295)."""
    p = tmp_path / "g.mtx"
    p.write_text("%%MatrixMarket matrix coordinate real general\n% c\n3 4 4\n3 1 5.0\n1 4 2.0\n1 2 7.0\n3 3 1.5\n")
    a = M.read_mtx(str(p), keep_values=True)
    assert (a.m, a.n, a.nnz) == (3, 4, 4)
    assert a.row_ptr.tolist() == [0, 2, 2, 4] and a.col.tolist() == [3, 1, 0, 2]
    assert a.val.tolist() == [2.0, 7.0, 5.0, 1.5]
    assert np.all(M.read_mtx(str(p)).val == 0)  # default: values discarded, caller fills rand()%10
    s = tmp_path / "s.mtx"
    s.write_text("%%MatrixMarket matrix coordinate pattern symmetric\n3 3 3\n2 1\n3 3\n3 1\n")
    b = M.read_mtx(str(s), keep_values=True)
    assert b.nnz == 5 and b.row_ptr.tolist() == [0, 2, 3, 5]
    assert b.col.tolist() == [1, 2, 0, 2, 0] and np.all(b.val == 1.0)
    c = tmp_path / "c.mtx"
    c.write_text("%%MatrixMarket matrix coordinate complex general\n1 1 1\n1 1 1.0 2.0\n")
    with pytest.raises(M.MtxError) as e:
        M.read_mtx(str(c))
    assert e.value.code == -3
    with pytest.raises(M.MtxError) as e:
        M.read_mtx(str(tmp_path / "missing.mtx"))
    assert e.value.code == -1
    bad = tmp_path / "bad.mtx"
    bad.write_text("hello\n")
    with pytest.raises(M.MtxError) as e:
        M.read_mtx(str(bad))
    assert e.value.code == -2
    mat = M.example_matrix()
    mat.val[:] = np.arange(mat.nnz) % 7
    M.write_mtx(str(tmp_path / "rt.mtx"), mat)
    back = M.read_mtx(str(tmp_path / "rt.mtx"), keep_values=True)
    assert np.array_equal(back.row_ptr, mat.row_ptr) and np.array_equal(back.col, mat.col)
    assert np.array_equal(back.val, mat.val)


def test_synthetic_workloads_have_the_catalogue_shape():
    sc = M.scircuit_like(scale=0.05)
    assert sc.nnz == int(958_936 * 0.05) and np.diff(sc.row_ptr).min() >= 1
    wb = M.webbase_like(scale=0.02)
    assert wb.nnz == int(3_105_536 * 0.02) and (np.diff(wb.row_ptr) == 0).mean() >= 0.10
    rm = M.rmat(10, 8)
    assert (rm.m, rm.nnz) == (1024, 8192)
    blk = M.rmat(10, 8, row_lo=256, row_hi=512)
    full_rows = np.repeat(np.arange(1024), np.diff(rm.row_ptr))
    sel = (full_rows >= 256) & (full_rows < 512)
    assert blk.nnz == sel.sum() and np.array_equal(np.sort(blk.col), np.sort(rm.col[sel]))


def test_nnz_balanced_row_partition():
    mat = M.webbase_like(scale=0.02)
    for parts in (1, 2, 3, 8):
        cuts = S.partition_rows_by_nnz(mat.row_ptr, parts)
        assert cuts[0] == 0 and cuts[-1] == mat.m and np.all(np.diff(cuts) >= 0)
        nnz_per = np.diff(mat.row_ptr[cuts].astype(np.int64))
        assert nnz_per.sum() == mat.nnz
        longest = int(np.diff(mat.row_ptr).max())
        assert nnz_per.max() <= mat.nnz / parts + longest  # balanced up to one row
        blocks = [S.extract_row_block(mat.row_ptr, mat.col, mat.val, mat.n, cuts, r) for r in range(parts)]
        assert np.array_equal(np.concatenate([b.col for b in blocks]), mat.col)
        assert all(b.row_ptr[0] == 0 and b.row_ptr[-1] == b.nnz for b in blocks)
        # cost balance (the default of every sharded path): non-zeros + ROW_WEIGHT per row, same guarantees
        for w in (S.ROW_WEIGHT, 5):
            cc = S.partition_rows_by_cost(mat.row_ptr, parts, w)
            assert cc[0] == 0 and cc[-1] == mat.m and np.all(np.diff(cc) >= 0)
            cost = np.diff(mat.row_ptr[cc].astype(np.int64)) + w * np.diff(cc)
            assert cost.sum() == mat.nnz + w * mat.m
            assert cost.max() <= (mat.nnz + w * mat.m) / parts + longest + w
        assert np.array_equal(S.partition_rows_by_cost(mat.row_ptr, parts, 0), cuts)


def _gloo_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle.csr5_oracle import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mat = M.webbase_like(scale=0.01, seed=3)
        val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=4, mode="int")
        orc = Oracle()
        sh = S.ShardedSpmv(mat.row_ptr, mat.col, val, mat.n, rank, world)
        # only rank 0 holds x before the broadcast -- the ONE collective of the path
        xt = torch.from_numpy(x.copy()) if rank == 0 else torch.zeros(mat.n, dtype=torch.float64)

        def local(block, xb):  # CPU stand-in for the per-rank HIP kernel: CSR5 at omega=64 via the oracle
            f = orc.convert(64, 4, block.m, block.row_ptr, block.col, block.val)
            return torch.from_numpy(orc.spmv(f, block.row_ptr, xb.numpy()))

        y_local = sh.run(xt, local)
        y = S.gather_y(y_local, sh.cuts).numpy()
        ref = orc.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
        q.put((rank, bool(np.array_equal(y, ref)), int(sh.block.nnz)))
    finally:
        dist.destroy_process_group()


def test_sharded_spmv_world2_gloo():
    """N > 1 path on CPU: cost-balanced row blocks, one broadcast of x, per-rank SpMV, y stays sharded."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29500 + (os.getpid() % 2000)
    procs = [ctx.Process(target=_gloo_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=180) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [True, True]
    total = M.webbase_like(scale=0.01, seed=3).nnz
    assert res[0][2] + res[1][2] == total and abs(res[0][2] - res[1][2]) < 0.2 * total


# ---------------------------------------------------------------------------------------------------
# coupled iterations: y shards written into the next x + in-place all-gather
# ---------------------------------------------------------------------------------------------------
def test_padded_layout_roundtrip_and_column_remap():
    rng = np.random.default_rng(0)
    for cuts in ([0, 10, 10, 37, 100], [0, 100], [0, 0, 64, 130], [0, 1, 2, 3]):
        lay = S.PaddedLayout(cuts)
        n = cuts[-1]
        assert lay.width % S.PaddedLayout.ALIGN == 0 and lay.width >= np.diff(cuts).max()
        v = rng.standard_normal(n)
        padded = lay.to_padded(v)
        assert padded.size == lay.padded_len and np.array_equal(lay.from_padded(padded), v)
        col = rng.integers(0, n, 500)
        assert np.array_equal(padded[lay.remap_columns(col)], v[col])  # the remapped gather reads the same x
        own = lay.owner(col)
        assert np.all((np.asarray(cuts)[own] <= col) & (col < np.asarray(cuts)[own + 1]))


def _coupled_problem():
    mat = M.rmat(scale=11, edge_factor=8, seed=5)            # square, skewed rows, empty rows
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, seed=6, mode="pos")
    return mat, val, x / np.linalg.norm(x)


def _power_reference(orc, mat, val, x0, iters):
    x, lam = x0.copy(), None
    for _ in range(iters):
        y = orc.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x)
        lam = float(np.dot(x, y) / np.dot(x, x))
        x = y / np.linalg.norm(y)
    return x, lam


def _coupled_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch
    import torch.distributed as dist
    from oracle.csr5_oracle import Oracle
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        mat, val, x0 = _coupled_problem()
        orc = Oracle()
        cp = S.CoupledSpmv(mat.row_ptr, mat.col, val, mat.n, rank, world)
        lay = cp.layout

        def local(block, xp, y_slot):  # CPU stand-in for the per-rank HIP kernel
            y = orc.csr_spmv(block.m, block.row_ptr, block.col, block.val, xp.numpy())
            y_slot.copy_(torch.from_numpy(y))

        a = torch.from_numpy(lay.to_padded(x0))
        b = torch.zeros_like(a)
        # one plain step first: every rank must end up with the complete A x
        cp.step(local, a, b)
        y1 = lay.from_padded(b.numpy())
        ok_step = bool(np.array_equal(y1, orc.csr_spmv(mat.m, mat.row_ptr, mat.col, val, x0)))
        b.zero_()
        xk, lam = cp.power_iteration(local, a, b, iters=12)
        x_ref, lam_ref = _power_reference(orc, mat, val, x0, 12)
        err = float(np.max(np.abs(lay.from_padded(xk.numpy()) - x_ref)))
        q.put((rank, ok_step, err, abs(float(lam) - lam_ref) / abs(lam_ref), int(cp.block.nnz)))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_coupled_power_iteration_gloo(world):
    """per-iteration collective path on CPU: y shard -> slot of the next x -> in-place all-gather."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 31500 + (os.getpid() % 2000) + world
    procs = [ctx.Process(target=_coupled_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert all(r[1] for r in res), "A x differs after one coupled step"
    assert max(r[2] for r in res) < 1e-12 and max(r[3] for r in res) < 1e-12
    assert sum(r[4] for r in res) == _coupled_problem()[0].nnz


def _build_c_host(out_dir) -> str:
    """gcc -std=c99 on tests/c/abi_smoke.c against include/csr5hip.h + libcsr5hip.so: the ABI really is plain C."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(str(out_dir), "abi_smoke")
    lib_dir = os.path.join(root, "benchmark_spmv_using_csr5_amd")
    subprocess.check_call(["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(root, "include"),
                           os.path.join(root, "tests", "c", "abi_smoke.c"), "-o", exe, "-L", lib_dir, "-lcsr5hip",
                           "-Wl,-rpath," + lib_dir, "-lm"])
    return exe


def test_plain_c_host_compiles_and_links(tmp_path):
    assert os.path.exists(_build_c_host(tmp_path))


def test_new_option_keys_and_multi_handle_argument_checks():
    """Host-side checks of the round-2 entry points that need no device: option ranges, multi-handle argument errors."""
    lib = _capi.load()
    h = C.c_void_p()
    assert lib.csr5hip_create(C.byref(h), 10, 10, _capi.F64) == 0
    assert lib.csr5hip_input_csr(h, 100, None, None, None) == 0
    for bad in (3, 5, 65, 128, -1):
        assert lib.csr5hip_set_option(h, _capi.OPT_COLUMN_SLABS, bad) == _capi.INVALID_ARGUMENT
    for ok in (0, 1, 2, 8, 64):
        assert lib.csr5hip_set_option(h, _capi.OPT_COLUMN_SLABS, ok) == 0
    assert lib.csr5hip_set_option(h, _capi.OPT_SLAB_SHIFT, 25) == _capi.INVALID_ARGUMENT
    assert lib.csr5hip_set_option(h, _capi.OPT_SLAB_SHIFT, 9) == 0
    assert lib.csr5hip_set_option(h, _capi.OPT_SLAB_HOT, 3) == _capi.INVALID_ARGUMENT
    assert lib.csr5hip_set_option(h, _capi.OPT_SLAB_HOT, 2) == 0
    assert lib.csr5hip_set_option(h, _capi.OPT_ZERO_EMPTY_ROWS, 1) == 0
    # round 4: how the hot-table kernel's permuted copy of x follows x (0 = per spmv, the reference's live-x semantics; 1 = per setX)
    assert lib.csr5hip_set_option(h, _capi.OPT_X_SNAPSHOT, 2) == _capi.INVALID_ARGUMENT
    assert lib.csr5hip_set_option(h, _capi.OPT_X_SNAPSHOT, 1) == 0
    info = _capi.Csr5Info()
    assert lib.csr5hip_get_info(h, C.byref(info)) == 0 and info.column_slabs == 0 and info.slab_hot == 0
    assert info.x_snapshot == 1 and info.slab_x_permuted == 0 and info.slab_cold_entries == 0
    assert lib.csr5hip_set_option(h, _capi.OPT_X_SNAPSHOT, 0) == 0
    assert lib.csr5hip_set_option(h, _capi.OPT_NARROW_VALUES, 2) == _capi.INVALID_ARGUMENT
    assert lib.csr5hip_set_option(h, _capi.OPT_NARROW_VALUES, 1) == 0  # (no conversion yet: only remembered)
    assert lib.csr5hip_set_option(h, _capi.OPT_NARROW_VALUES, 0) == 0
    # the Python constants are the header's
    hdr = open(os.path.join(ROOT, "include", "csr5hip.h")).read()
    for name, val in (("CSR5HIP_OPT_NARROW_VALUES", _capi.OPT_NARROW_VALUES), ("CSR5HIP_OPT_X_SNAPSHOT", _capi.OPT_X_SNAPSHOT), ("CSR5HIP_MULTI_OPT_OWN_REPLICAS", _capi.MULTI_OPT_OWN_REPLICAS),
                      ("CSR5HIP_OPT_SLAB_MEMORY_MIB", _capi.OPT_SLAB_MEMORY_MIB)):
        assert f"#define {name} {val}" in " ".join(hdr.split()), name
    assert lib.csr5hip_spmv_repeat(h, 1.0, C.c_void_p(8), 3) == _capi.UNSUPPORTED_CSR_SPMV
    assert lib.csr5hip_free(h) == 0
    mh = C.c_void_p()
    devs = (C.c_int * 2)(0, 0)
    assert lib.csr5hip_multi_create(C.byref(mh), devs, 0, 10, 10, _capi.F64) == _capi.INVALID_ARGUMENT
    assert lib.csr5hip_multi_create(C.byref(mh), devs, 2, 10, 10, 9) == _capi.UNSUPPORTED_VALUE_TYPE
    assert lib.csr5hip_multi_spmv(None, 1.0) == _capi.INVALID_ARGUMENT
    assert lib.csr5hip_spmv_rotate(None, None, 2, 1.0, 4) == _capi.INVALID_ARGUMENT


def test_rmat_shards_are_row_blocks_of_one_matrix():
    """Strong scaling input (BASELINE config 3): the per-shard generator gives every world size the SAME global matrix;
    the shards are its cost-balanced row blocks (sharding.partition_rows_by_cost; row_weight = 0: by nnz), generated
    without materialising it."""
    torch = pytest.importorskip("torch")
    full = M.rmat_device_shard(11, 8, seed=3, rank=0, world=1, device="cpu", chunk_log2=12)
    rp = full.row_ptr.numpy().astype(np.int64)
    col = full.col.numpy()
    assert full.m == 1 << 11 and full.nnz == (1 << 11) * 8 and rp[-1] == full.nnz
    for world, weight in ((2, None), (3, 0), (8, None), (8, 7)):
        cuts = S.partition_rows_by_cost(rp, world, S.ROW_WEIGHT if weight is None else weight)
        lo_rows = 0
        for rank in range(world):
            sh = M.rmat_device_shard(11, 8, seed=3, rank=rank, world=world, device="cpu", chunk_log2=12, row_weight=weight)
            lo, hi = int(cuts[rank]), int(cuts[rank + 1])
            assert sh.m == hi - lo and sh.n == full.n
            assert np.array_equal(sh.row_ptr.numpy().astype(np.int64), rp[lo:hi + 1] - rp[lo])
            assert np.array_equal(sh.col.numpy(), col[rp[lo]:rp[hi]])
            lo_rows += sh.m
        assert lo_rows == full.m

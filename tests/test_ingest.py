"""Matrix Market ingest + COO->CSR.

CPU part (-m "not gpu"):
  * the oracle's restatement of the reference CLI ingest (oracle/mtx_oracle.c) against the CSR the
    REFERENCE itself built from the committed fixtures (tests/golden/mtx/expected.npz, produced by
    oracle/gen_golden_mtx.py through oracle/_ref/libref_ingest.so), and against the live reference
    when oracle/_ref is present;
  * the library's multi-threaded host parser (csr5hip_mtx_read, no GPU involved) against the same
    fixtures and, on a generated 200k-entry file, against the oracle for several thread counts.
GPU part (-m gpu): csr5hip_coo_to_csr / csr5hip_mtx_load against the oracle, bit for bit
(row_ptr, col_idx, val), plus size-independent properties on a large COO.
"""
import glob
import os

import numpy as np
import pytest

from benchmark_spmv_using_csr5_amd import _capi, ingest
from benchmark_spmv_using_csr5_amd.matrices import MtxError
from oracle.csr5_oracle import MtxExit, Oracle, Reference

MTX_DIR = os.path.join(os.path.dirname(__file__), "golden", "mtx")
EXPECTED = np.load(os.path.join(MTX_DIR, "expected.npz"))
FIXTURES = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(MTX_DIR, "*.mtx")))
GOOD = [f for f in FIXTURES if int(EXPECTED[f + ".code"]) == 0]
BAD = [f for f in FIXTURES if int(EXPECTED[f + ".code"]) != 0]


def path_of(name):
    return os.path.join(MTX_DIR, name + ".mtx")


def expected_csr(name):
    m, n = (int(v) for v in EXPECTED[name + ".dims"])
    return m, n, EXPECTED[name + ".row_ptr"], EXPECTED[name + ".col"], EXPECTED[name + ".val"]


def assert_same_values(a, b):
    """bit-for-bit including the sign of zero and denormals"""
    assert np.array_equal(np.asarray(a, dtype=np.float64).view(np.uint64),
                          np.asarray(b, dtype=np.float64).view(np.uint64))


def test_fixture_set_is_complete():
    assert len(GOOD) >= 15 and len(BAD) >= 5
    assert {int(EXPECTED[f + ".code"]) for f in BAD} == {-2, -3, -4}


# ---------------------------------------------------------------------------------------------
# oracle vs reference goldens
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("name", GOOD)
def test_oracle_ingest_matches_reference_golden(oracle, name):
    m, n, row_ptr, col, val = expected_csr(name)
    got = oracle.mtx_read(path_of(name))
    assert (got.m, got.n, got.nnz) == (m, n, col.size)
    assert np.array_equal(got.row_ptr, row_ptr)
    assert np.array_equal(got.col, col)
    assert_same_values(got.val, val)


@pytest.mark.parametrize("name", BAD)
def test_oracle_ingest_exit_codes(oracle, name):
    with pytest.raises(MtxExit) as e:
        oracle.mtx_read(path_of(name))
    assert e.value.code == int(EXPECTED[name + ".code"])


def test_oracle_missing_file(oracle, tmp_path):
    with pytest.raises(MtxExit) as e:
        oracle.mtx_read(str(tmp_path / "nope.mtx"))
    assert e.value.code == -1


@pytest.mark.skipif(not Reference.ingest_available(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("name", GOOD)
def test_oracle_ingest_matches_live_reference(oracle, name):
    m, n, row_ptr, col, val = Reference().ingest(path_of(name))
    got = oracle.mtx_read(path_of(name))
    assert (got.m, got.n) == (m, n)
    assert np.array_equal(got.row_ptr, row_ptr) and np.array_equal(got.col, col)
    assert_same_values(got.val, val)


@pytest.mark.skipif(not Reference.ingest_available(), reason="oracle/_ref not built (no /root/reference)")
@pytest.mark.parametrize("name", [f for f in GOOD if f not in ("empty_matrix",)])
def test_reference_cli_runs_on_the_fixtures(name):
    """BASELINE.json configs[0] (plumbing + correctness baseline on the host CPU, no GPU): the reference's own CLI
    -- ingest, CSR5_avx2 conversion, SpMV loop, self-check -- compiled from its sources where they lie, accepts
    every fixture, reports the dimensions the goldens hold and passes its own check."""
    m, n, row_ptr, col, val = expected_csr(name)
    rc, text = Reference().cli(path_of(name))
    assert rc == 0, text
    assert f" ( {m}, {n} ) nnz = {col.size}\n" in text
    assert "omega = 4, sigma = 16." in text and "Check... PASS!" in text


# ---------------------------------------------------------------------------------------------
# the library's host parser (no GPU needed)
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("threads", [1, 3])
@pytest.mark.parametrize("name", GOOD)
def test_native_parser_matches_reference_golden(oracle, name, threads):
    """COO from the parallel parser, pushed through the reference's scatter order (oracle), must give the
    reference's CSR; the triplets themselves must equal the fscanf scanner's."""
    m, n, row_ptr, col, val = expected_csr(name)
    coo = ingest.read_mtx_coo(path_of(name), threads=threads)
    seq = oracle.mtx_read(path_of(name))
    assert (coo.m, coo.n, coo.nz, coo.symmetric, coo.field) == (m, n, seq.nz_file, seq.symmetric, seq.field)
    assert np.array_equal(coo.row, seq.coo_row) and np.array_equal(coo.col, seq.coo_col)
    assert_same_values(coo.val, seq.coo_val)
    rp, ci, cv = oracle.coo_to_csr(m, coo.row, coo.col, coo.val, coo.symmetric)
    assert np.array_equal(rp, row_ptr) and np.array_equal(ci, col)
    assert_same_values(cv, val)
    assert coo.fast_path == (name != "free_form_tokens")


@pytest.mark.parametrize("name", BAD)
def test_native_parser_exit_codes(name):
    with pytest.raises(MtxError) as e:
        ingest.read_mtx_coo(path_of(name))
    assert e.value.code == int(EXPECTED[name + ".code"])


def test_native_parser_missing_truncated_and_out_of_range(tmp_path):
    with pytest.raises(MtxError) as e:
        ingest.read_mtx_coo(str(tmp_path / "nope.mtx"))
    assert e.value.code == -1
    short = tmp_path / "short.mtx"
    short.write_text("%%MatrixMarket matrix coordinate real general\n3 3 4\n1 1 1\n2 2 2\n")
    with pytest.raises(MtxError) as e:  # the reference would carry on with uninitialised indices
        ingest.read_mtx_coo(str(short))
    assert e.value.code == -4
    oob = tmp_path / "oob.mtx"
    oob.write_text("%%MatrixMarket matrix coordinate real general\n3 3 2\n1 1 1\n4 2 2\n")
    with pytest.raises(ValueError):     # the reference would write outside its arrays
        ingest.read_mtx_coo(str(oob))
    zero = tmp_path / "zero.mtx"
    zero.write_text("%%MatrixMarket matrix coordinate pattern symmetric\n3 3 1\n0 1\n")
    with pytest.raises(ValueError):
        ingest.read_mtx_coo(str(zero))


def write_big(path, rng, m, n, nz, field, symmetric):
    r = rng.integers(1, m + 1, nz)
    c = rng.integers(1, n + 1, nz)
    if symmetric:
        r, c = np.maximum(r, c), np.minimum(r, c)
    head = f"%%MatrixMarket matrix coordinate {field} {'symmetric' if symmetric else 'general'}\n% generated\n{m} {n} {nz}\n"
    if field == "pattern":
        body = "\n".join(f"{a} {b}" for a, b in zip(r, c))
    elif field == "integer":
        iv = rng.integers(-1000, 1000, nz)
        body = "\n".join(f"{a} {b} {x}" for a, b, x in zip(r, c, iv))
    else:
        v = rng.standard_normal(nz) * 10.0 ** rng.integers(-30, 30, nz)
        body = "\n".join(f"{a} {b} {float(x)!r}" if k % 3 else f"{a} {b} {x:.9e}" for k, (a, b, x) in enumerate(zip(r, c, v)))
    with open(path, "w") as f:
        f.write(head + body + "\n")


@pytest.mark.parametrize("field,symmetric", [("real", False), ("real", True), ("pattern", True), ("integer", False)])
def test_native_parser_parallel_equals_sequential_scanner(oracle, tmp_path, field, symmetric):
    rng = np.random.default_rng(7)
    p = str(tmp_path / "big.mtx")
    write_big(p, rng, 5000, 5000, 200_000, field, symmetric)
    seq = oracle.mtx_read(p)
    for threads in (1, 2, 7, 16):
        coo = ingest.read_mtx_coo(p, threads=threads)
        assert coo.fast_path and coo.threads == threads
        assert np.array_equal(coo.row, seq.coo_row) and np.array_equal(coo.col, seq.coo_col)
        assert_same_values(coo.val, seq.coo_val)


# ---------------------------------------------------------------------------------------------
# device COO -> CSR
# ---------------------------------------------------------------------------------------------
def device_csr_of(coo_row, coo_col, coo_val, m, n, symmetric, dtype=np.float64):
    import torch
    dev = torch.device("cuda:0")
    r = torch.from_numpy(np.ascontiguousarray(coo_row, dtype=np.int32)).to(dev)
    c = torch.from_numpy(np.ascontiguousarray(coo_col, dtype=np.int32)).to(dev)
    v = None if coo_val is None else torch.from_numpy(np.ascontiguousarray(coo_val, dtype=np.float64)).to(dev)
    d = ingest.coo_to_csr(m, n, r, c, v, symmetric, dtype=dtype)
    host = d.to_host()
    has_val = d.val != 0
    d.release()
    return host, has_val


@pytest.mark.gpu
@pytest.mark.parametrize("name", GOOD)
def test_gpu_mtx_load_matches_reference_golden(name):
    m, n, row_ptr, col, val = expected_csr(name)
    d = ingest.load_mtx(path_of(name), dtype=np.float64, threads=2)
    host = d.to_host()
    assert (d.m, d.n, d.nnz) == (m, n, col.size)
    assert np.array_equal(host.row_ptr, row_ptr) and np.array_equal(host.col, col)
    assert_same_values(host.val, val)
    d.release()
    d32 = ingest.load_mtx(path_of(name), dtype=np.float32)
    h32 = d32.to_host()
    assert h32.val.dtype == np.float32
    with np.errstate(over="ignore"):
        assert np.array_equal(h32.val.view(np.uint32), val.astype(np.float32).view(np.uint32))
    d32.release()


@pytest.mark.gpu
@pytest.mark.parametrize("name", BAD)
def test_gpu_mtx_load_exit_codes(name):
    with pytest.raises(MtxError) as e:
        ingest.load_mtx(path_of(name))
    assert e.value.code == int(EXPECTED[name + ".code"])


@pytest.mark.gpu
@pytest.mark.parametrize("symmetric", [False, True])
@pytest.mark.parametrize("m,n,nz", [(1, 1, 1), (7, 7, 0), (1000, 1000, 50_000), (200_000, 200_000, 300_000),
                                    (50, 50, 400_000), (3, 100_000, 100_000)])
def test_gpu_coo_to_csr_matches_oracle(oracle, m, n, nz, symmetric):
    if symmetric and m != n:
        n = m
    rng = np.random.default_rng(m * 31 + nz)
    row = rng.integers(0, m, nz).astype(np.int32)
    col = rng.integers(0, n, nz).astype(np.int32)
    val = rng.standard_normal(nz)
    rp, ci, cv = oracle.coo_to_csr(m, row, col, val, symmetric)
    host, has_val = device_csr_of(row, col, val, m, n, symmetric)
    assert has_val or nz == 0
    assert np.array_equal(host.row_ptr, rp) and np.array_equal(host.col, ci)
    assert_same_values(host.val, cv)
    # structure only
    host2, has_val2 = device_csr_of(row, col, None, m, n, symmetric)
    assert not has_val2
    assert np.array_equal(host2.row_ptr, rp) and np.array_equal(host2.col, ci)


@pytest.mark.gpu
def test_gpu_coo_to_csr_rejects_out_of_range():
    row = np.array([0, 5], dtype=np.int32)
    col = np.array([0, 1], dtype=np.int32)
    with pytest.raises(ValueError):
        device_csr_of(row, col, np.ones(2), 3, 3, False)
    with pytest.raises(ValueError):  # the mirror of (0, 4) would land in row 4 of a 3-row matrix
        device_csr_of(np.array([0], np.int32), np.array([4], np.int32), np.ones(1), 3, 6, True)


@pytest.mark.gpu
def test_gpu_coo_to_csr_large_properties():
    """16 M entries, symmetric: stable-sort properties that do not need the oracle (size-independent)."""
    import torch
    dev = torch.device("cuda:0")
    g = torch.Generator(device=dev).manual_seed(5)
    m = 1 << 20
    nz = 1 << 24
    r = torch.randint(0, m, (nz,), generator=g, device=dev, dtype=torch.int32)
    c = torch.randint(0, m, (nz,), generator=g, device=dev, dtype=torch.int32)
    v = torch.arange(nz, device=dev, dtype=torch.float64)  # the value names the entry
    d = ingest.coo_to_csr(m, m, r, c, v, True)
    off = int((r != c).sum())
    assert d.nnz == nz + off
    host = d.to_host()
    d.release()
    rp = host.row_ptr.astype(np.int64)
    assert rp[0] == 0 and rp[-1] == host.nnz and np.all(np.diff(rp) >= 0)
    counts = (torch.bincount(r.long(), minlength=m) + torch.bincount(c[r != c].long(), minlength=m)).cpu().numpy()
    assert np.array_equal(np.diff(rp), counts)
    # inside every row the entry numbers (values) never decrease: file order survived
    rows = np.repeat(np.arange(m), np.diff(rp))
    same_row = rows[1:] == rows[:-1]
    assert np.all(host.val[1:][same_row] >= host.val[:-1][same_row])
    # every CSR element is its entry or the mirror of its entry
    e = host.val.astype(np.int64)
    rr, cc = r.cpu().numpy(), c.cpu().numpy()
    direct = (rr[e] == rows) & (cc[e] == host.col)
    mirror = (cc[e] == rows) & (rr[e] == host.col)
    assert np.all(direct | mirror)
    assert int(direct.sum()) + int((mirror & ~direct).sum()) == host.nnz


@pytest.mark.gpu
def test_gpu_ingest_feeds_the_handle(oracle, tmp_path):
    """load_mtx -> inputCSR -> asCSR5 -> spmv == the oracle on the reference-ordered CSR."""
    import torch
    from benchmark_spmv_using_csr5_amd.handle import anonymouslibHandle
    rng = np.random.default_rng(11)
    p = str(tmp_path / "sym.mtx")
    write_big(p, rng, 3000, 3000, 40_000, "integer", True)
    d = ingest.load_mtx(p)
    seq = oracle.mtx_read(p)
    x = rng.integers(0, 10, d.n).astype(np.float64)
    dev = torch.device("cuda:0")
    xd = torch.from_numpy(x).to(dev)
    yd = torch.zeros(d.m, dtype=torch.float64, device=dev)
    A = anonymouslibHandle(d.m, d.n)
    assert A.inputCSR(d.nnz, d.row_ptr, d.col_idx, d.val) == 0
    A.setX(xd)
    A.setSigma(_capi.AUTO_TUNED_SIGMA)
    assert A.asCSR5() == 0
    assert A.spmv(1.0, yd) == 0
    torch.cuda.synchronize()
    y_ref = oracle.csr_spmv(seq.m, seq.row_ptr, seq.col, seq.val, x)
    assert np.array_equal(yd.cpu().numpy(), y_ref)  # integer data: exact
    A.destroy()
    A.close()
    d.release()


@pytest.mark.gpu
def test_gpu_cli_on_every_fixture():
    """`./spmv fixture.mtx`: the ` ( m, n ) nnz = ` line must carry the reference's numbers (symmetric
    expansion included), the self-check must pass, broken files must give the reference's exit codes."""
    import subprocess
    exe = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                       "benchmark_spmv_using_csr5_amd", "csrc", "spmv")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    exe32 = exe + "_f32"   # the reference builds one binary per precision (make VALUE_TYPE=float)
    assert os.path.exists(exe32), "run __graft_entry__.build() first"
    for name in GOOD:
        m, n, row_ptr, col, val = expected_csr(name)
        if col.size == 0:
            continue  # the reference CLI divides by zero tiles on an empty matrix; nothing to compare
        for binary, banner in ((exe, "64-bit Double Precision"), (exe32, "32-bit Single Precision")):
            out = subprocess.run([binary, path_of(name)], capture_output=True, text=True, timeout=300,
                                 env=dict(os.environ, CSR5_SEED="3"))
            assert out.returncode == 0, (name, out.stderr)
            assert f"PRECISION = {banner}" in out.stdout
            assert f" ( {m}, {n} ) nnz = {col.size}\n" in out.stdout, (name, out.stdout)
            assert "Check... PASS!" in out.stdout, (name, out.stdout)
    for name in BAD:
        rc = subprocess.run([exe, path_of(name)], capture_output=True, timeout=60).returncode
        assert rc == 256 + int(EXPECTED[name + ".code"]), name

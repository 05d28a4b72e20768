"""Matrix inputs for the CSR5 hot path: Matrix Market ingest and seeded synthetic stand-ins.

The reference CLI reads a Matrix Market coordinate file, expands symmetric/hermitian storage, builds CSR
by a counting sort that keeps FILE order inside each row (duplicates kept), then DISCARDS the file's
values and fills matrix and x with ``rand() % 10`` (CSR5_avx2/main.cpp:135-295).  ``read_mtx`` follows
that contract; the fill is seeded here (the reference seeds with ``time(NULL)``).

SuiteSparse files are not available offline, so ``scircuit_like`` / ``webbase_like`` / ``nd24k_like`` /
``rmat`` generate seeded stand-ins with the catalogue dimensions and row-length character of the
BASELINE.json configs.  They are labelled synthetic wherever they are reported.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np


@dataclass
class CsrMatrix:
    m: int
    n: int
    row_ptr: np.ndarray  # int32[m+1]
    col: np.ndarray      # int32[nnz]
    val: np.ndarray      # float64/float32[nnz]
    name: str = "csr"

    @property
    def nnz(self) -> int:
        return int(self.row_ptr[self.m])


def algorithmic_bytes(m: int, n: int, nnz: int, value_size: int) -> int:
    """B_alg of SURVEY.md section 8(d): every CSR array, x and y moved exactly once."""
    return nnz * (4 + value_size) + 4 * (m + 1) + value_size * (n + m)


def reference_getB(m: int, nnz: int, value_size: int) -> int:
    """The reference CLI's own byte count (detail/utils.h:10-14); CLI parity only."""
    return (m + 1 + nnz) * 4 + (2 * nnz + m) * value_size


def fill_values(nnz: int, n: int, dtype, seed: int, mode: str = "int"):
    """Matrix values and x.  mode 'int' = rand()%10 as CSR5_avx2/main.cpp:286-295 (exact in fp);
    'real' = uniform(-1, 1); 'pos' = uniform(0.1, 1) (no cancellation: strict relative checks)."""
    rng = np.random.default_rng(seed)
    if mode == "int":
        val = rng.integers(0, 10, size=nnz).astype(dtype)
        x = rng.integers(0, 10, size=n).astype(dtype)
    elif mode == "real":
        val = rng.uniform(-1.0, 1.0, size=nnz).astype(dtype)
        x = rng.uniform(-1.0, 1.0, size=n).astype(dtype)
    elif mode == "pos":
        val = rng.uniform(0.1, 1.0, size=nnz).astype(dtype)
        x = rng.uniform(0.1, 1.0, size=n).astype(dtype)
    else:
        raise ValueError(mode)
    return val, x


def csr_from_row_lengths(lengths, n: int, rng, band: float = 0.0, name="synthetic",
                         dtype=np.float64, far: str = "uniform") -> CsrMatrix:
    """CSR with the given row lengths; columns uniform over [0,n) or, with probability `band`,
    within +-64 of the diagonal.  Column order inside a row is NOT sorted (the reference keeps file
    order, main.cpp:266-275) and duplicates may occur.  `far` = how the columns that are NOT near the
    diagonal are drawn: "uniform" (default) or "powerlaw" (in-degree-like: column popularity ~ rank^-1 over a
    random relabelling of the columns, the locality-axis variant of scripts/experiments/round6/locality.py)."""
    lengths = np.asarray(lengths, dtype=np.int64)
    m = lengths.size
    row_ptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lengths, out=row_ptr[1:])
    nnz = int(row_ptr[m])
    assert nnz < 2**31
    if far == "powerlaw" and nnz:  # popularity ~ 1 / rank: inverse-cdf sampling of a log-uniform rank, hubs relabelled at random
        rank = np.minimum(np.floor(np.exp(rng.random(nnz) * np.log(n))).astype(np.int64) - 1, n - 1)
        col = rng.permutation(n)[np.maximum(rank, 0)]
    else:
        col = rng.integers(0, n, size=nnz, dtype=np.int64)
    if band > 0.0 and nnz:
        rows = np.repeat(np.arange(m, dtype=np.int64), lengths)
        near = rng.random(nnz) < band
        diag = (rows * n) // max(m, 1)
        col = np.where(near, np.clip(diag + rng.integers(-64, 65, size=nnz), 0, n - 1), col)
    return CsrMatrix(m, n, row_ptr.astype(np.int32), col.astype(np.int32),
                     np.zeros(nnz, dtype=dtype), name)


def _scale_to_total(lengths: np.ndarray, total: int, rng, cap: int) -> np.ndarray:
    """Adjust integer row lengths so they sum to `total` exactly (keeps zeros zero where possible)."""
    lengths = np.minimum(lengths, cap).astype(np.int64)
    diff = total - int(lengths.sum())
    nz = np.flatnonzero(lengths > 0)
    while diff != 0:
        k = min(abs(diff), nz.size)
        pick = rng.choice(nz, size=k, replace=False)
        if diff > 0:
            lengths[pick] += 1
            lengths[pick] = np.minimum(lengths[pick], cap)
        else:
            lengths[pick] = np.maximum(lengths[pick] - 1, 1)
        diff = total - int(lengths.sum())
    return lengths


def scircuit_like(seed: int = 1, scale: float = 1.0, dtype=np.float64, row_cap: int = 353,
                  band: float = 0.5, far: str = "uniform") -> CsrMatrix:
    """SuiteSparse scircuit stand-in: 170 998 x 170 998, 958 936 nnz, heavy-tailed rows (mean 5.6,
    max ~350), no structural empty rows, half the entries near the diagonal."""
    rng = np.random.default_rng(seed)
    m = max(int(170_998 * scale), 16)
    nnz = max(int(958_936 * scale), m)
    raw = 1 + np.floor(rng.pareto(2.2, size=m) * 3.0).astype(np.int64)
    raw = _scale_to_total(raw, nnz, rng, cap=row_cap)
    name = "scircuit-like(synthetic)" if band == 0.5 and far == "uniform" else f"scircuit-like(synthetic, band={band:g}, far={far})"
    return csr_from_row_lengths(raw, m, rng, band=band, name=name, dtype=dtype, far=far)


def webbase_like(seed: int = 2, scale: float = 1.0, dtype=np.float64, band: float = 0.3, far: str = "uniform") -> CsrMatrix:
    """SuiteSparse webbase-1M stand-in: 1 000 005 square, 3 105 536 nnz, power-law rows capped at
    4 700, >= 10 % empty rows (stresses the empty-row offsets and the segmented sum).  `band` = share of
    the links that stay within +-64 of the diagonal (default 0.3: a harsh guess, 70 % of the x gathers are
    uniformly random; crawl-ordered web graphs keep most links inside a host, i.e. near the diagonal)."""
    rng = np.random.default_rng(seed)
    m = max(int(1_000_005 * scale), 16)
    nnz = max(int(3_105_536 * scale), m // 2)
    raw = np.floor(rng.pareto(1.6, size=m) * 1.6 + 1.0).astype(np.int64)
    raw[rng.random(m) < 0.12] = 0
    raw = _scale_to_total(raw, nnz, rng, cap=4700)
    name = "webbase-1M-like(synthetic)" if band == 0.3 and far == "uniform" else f"webbase-1M-like(synthetic, band={band:g}, far={far})"
    return csr_from_row_lengths(raw, m, rng, band=band, name=name, dtype=dtype, far=far)


def nd24k_like(seed: int = 3, scale: float = 1.0, dtype=np.float32) -> CsrMatrix:
    """SuiteSparse nd24k stand-in: 72 000 square, ~399 nnz/row, banded blocks (dense-ish rows)."""
    rng = np.random.default_rng(seed)
    m = max(int(72_000 * scale), 64)
    per_row = 399
    lengths = np.full(m, per_row, dtype=np.int64)
    row_ptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(lengths, out=row_ptr[1:])
    nnz = int(row_ptr[m])
    rows = np.repeat(np.arange(m, dtype=np.int64), lengths)
    off = np.tile(np.arange(per_row, dtype=np.int64) - per_row // 2, m)
    jitter = rng.integers(-2000, 2001, size=nnz) * (rng.random(nnz) < 0.1)
    col = np.clip(rows + off + jitter, 0, m - 1)
    return CsrMatrix(m, m, row_ptr.astype(np.int32), col.astype(np.int32),
                     np.zeros(nnz, dtype=dtype), "nd24k-like(synthetic)")


def rmat(scale: int, edge_factor: int = 16, seed: int = 4, row_lo: int = 0, row_hi: int | None = None,
         abcd=(0.57, 0.19, 0.19, 0.05), dtype=np.float64) -> CsrMatrix:
    """R-MAT (Graph500 parameters), duplicates kept.  Returns the row block [row_lo, row_hi) with
    global column indices and a rebased row_ptr (the shard a rank owns).
    numpy implementation: meant for scale <= ~20; larger scales are generated on the device by
    bench.py."""
    n = 1 << scale
    row_hi = n if row_hi is None else row_hi
    rng = np.random.default_rng(seed)
    ne = n * edge_factor
    a, b, c, _ = abcd
    rows = np.zeros(ne, dtype=np.int64)
    cols = np.zeros(ne, dtype=np.int64)
    for _lvl in range(scale):
        r = rng.random(ne)
        rbit = r >= (a + b)
        cbit = ((r >= a) & (r < a + b)) | (r >= a + b + c)
        rows = (rows << 1) | rbit
        cols = (cols << 1) | cbit
    keep = (rows >= row_lo) & (rows < row_hi)
    rows = rows[keep] - row_lo
    cols = cols[keep]
    order = np.argsort(rows, kind="stable")
    rows = rows[order]
    cols = cols[order]
    mloc = row_hi - row_lo
    counts = np.bincount(rows, minlength=mloc)
    row_ptr = np.zeros(mloc + 1, dtype=np.int64)
    np.cumsum(counts, out=row_ptr[1:])
    return CsrMatrix(mloc, n, row_ptr.astype(np.int32), cols.astype(np.int32),
                     np.zeros(cols.size, dtype=dtype), f"rmat{scale}(synthetic)")


def example_matrix(seed: int = 7, dtype=np.float64) -> CsrMatrix:
    """Stand-in for the README's absent example.mtx: 500 x 500, ~5 k nnz, rows 10-29 empty, one dense
    row."""
    rng = np.random.default_rng(seed)
    m = 500
    lengths = rng.integers(2, 18, size=m).astype(np.int64)
    lengths[10:30] = 0
    lengths[200] = 500
    return csr_from_row_lengths(lengths, m, rng, band=0.0, name="example(synthetic)", dtype=dtype)


# ---------------------------------------------------------------------------------------------
# Matrix Market ingest / writer (CSR5_avx2/main.cpp:135-275)
# ---------------------------------------------------------------------------------------------
class MtxError(Exception):
    def __init__(self, code: int, msg: str):
        super().__init__(msg)
        self.code = code  # the reference CLI's exit codes: -1 open, -2 banner, -3 complex, -4 size


def read_mtx(path: str, dtype=np.float64, keep_values: bool = False) -> CsrMatrix:
    try:
        f = open(path, "r")
    except OSError as e:
        raise MtxError(-1, str(e))
    with f:
        banner = f.readline().split()
        if len(banner) < 5 or banner[0] != "%%MatrixMarket" or banner[1].lower() != "matrix":
            raise MtxError(-2, "Could not process Matrix Market banner.")
        fmt, field, symm = banner[2].lower(), banner[3].lower(), banner[4].lower()
        if fmt != "coordinate":
            raise MtxError(-2, "Could not process Matrix Market banner.")
        if field == "complex":
            raise MtxError(-3, "Sorry, data type 'COMPLEX' is not supported. ")
        line = f.readline()
        while line and (line.startswith("%") or not line.strip()):
            line = f.readline()
        try:
            m, n, nz = (int(t) for t in line.split()[:3])
        except Exception:
            raise MtxError(-4, "bad size line")
        data = np.loadtxt(f, ndmin=2, dtype=np.float64, max_rows=nz) if nz else np.zeros((0, 3))
    ri = data[:, 0].astype(np.int64) - 1
    ci = data[:, 1].astype(np.int64) - 1
    vv = data[:, 2] if (field != "pattern" and data.shape[1] > 2) else np.ones(ri.size)
    if symm in ("symmetric", "hermitian"):
        # mirror off-diagonals right after each entry, as the scatter loop at main.cpp:241-265 does
        off = ri != ci
        reps = np.where(off, 2, 1)
        idx = np.repeat(np.arange(ri.size), reps)
        second = np.zeros(idx.size, dtype=bool)
        second[1:] = idx[1:] == idx[:-1]
        r2 = np.where(second, ci[idx], ri[idx])
        c2 = np.where(second, ri[idx], ci[idx])
        ri, ci, vv = r2, c2, vv[idx]
    order = np.argsort(ri, kind="stable")  # counting sort by row keeps file order inside a row
    counts = np.bincount(ri, minlength=m)
    row_ptr = np.zeros(m + 1, dtype=np.int64)
    np.cumsum(counts, out=row_ptr[1:])
    val = vv[order].astype(dtype) if keep_values else np.zeros(ri.size, dtype=dtype)
    return CsrMatrix(m, n, row_ptr.astype(np.int32), ci[order].astype(np.int32), val,
                     name=path.rsplit("/", 1)[-1])


def write_mtx(path: str, mat: CsrMatrix, field: str = "real", symmetric: bool = False) -> None:
    rows = np.repeat(np.arange(mat.m), np.diff(mat.row_ptr))
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate {field} {'symmetric' if symmetric else 'general'}\n")
        f.write(f"{mat.m} {mat.n} {mat.nnz}\n")
        for r, c, v in zip(rows, mat.col, mat.val):
            if field == "pattern":
                f.write(f"{r + 1} {c + 1}\n")
            elif field == "integer":
                f.write(f"{r + 1} {c + 1} {int(v)}\n")
            else:
                f.write(f"{r + 1} {c + 1} {float(v)!r}\n")


# ---------------------------------------------------------------------------------------------
# Device-side R-MAT (torch is plumbing here: RNG, sort and prefix sum on the GPU) for scales that
# numpy cannot generate in bench time (scale 24 = 268 M non-zeros).
# ---------------------------------------------------------------------------------------------
@dataclass
class DeviceCsr:
    m: int
    n: int
    nnz: int
    row_ptr: object  # torch.int32[m+1] on the device
    col: object      # torch.int32[nnz] on the device
    name: str = "rmat(device)"


def rmat_device(scale: int, edge_factor: int, seed: int, rank: int, world: int, device, dtype=None,
                abcd=(0.57, 0.19, 0.19, 0.05), strong: bool = False) -> DeviceCsr:
    """weak (default): one row block per rank, every rank owns a 2^scale-row R-MAT block whose columns
    are spread over the world * 2^scale global columns.
    strong: ONE global 2^scale R-MAT (same seed on every rank), cut into `world` cost-balanced row
    blocks (sharding.partition_rows_by_cost); the rank keeps its block with global columns."""
    import torch

    if strong and world > 1:
        from .sharding import partition_rows_by_cost
        full = rmat_device(scale, edge_factor, seed, 0, 1, device, abcd=abcd)
        cuts = partition_rows_by_cost(full.row_ptr.cpu().numpy(), world)
        lo, hi = int(cuts[rank]), int(cuts[rank + 1])
        a, b = int(full.row_ptr[lo]), int(full.row_ptr[hi])
        row_ptr = (full.row_ptr[lo: hi + 1] - a).to(torch.int32).contiguous()
        col = full.col[a:b].clone()
        n = full.n
        del full
        return DeviceCsr(hi - lo, n, b - a, row_ptr, col, f"rmat{scale}(synthetic) rows {lo}:{hi}")

    n_local = 1 << scale
    ne = n_local * edge_factor
    g = torch.Generator(device=device).manual_seed(seed + 1009 * rank)
    a, b, c, _ = abcd
    rows = torch.zeros(ne, dtype=torch.int64, device=device)
    cols = torch.zeros(ne, dtype=torch.int64, device=device)
    for _lvl in range(scale):
        r = torch.rand(ne, generator=g, device=device)
        rbit = (r >= (a + b)).to(torch.int64)
        cbit = (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int64)
        rows = (rows << 1) | rbit
        cols = (cols << 1) | cbit
        del r, rbit, cbit
    if world > 1:
        blk = torch.randint(0, world, (ne,), generator=g, device=device, dtype=torch.int64)
        cols = cols + blk * n_local
        del blk
    rows, order = torch.sort(rows, stable=True)
    cols = cols[order].to(torch.int32)
    del order
    counts = torch.bincount(rows, minlength=n_local)
    row_ptr = torch.zeros(n_local + 1, dtype=torch.int64, device=device)
    row_ptr[1:] = torch.cumsum(counts, 0)
    return DeviceCsr(n_local, n_local * world, int(ne), row_ptr.to(torch.int32), cols.contiguous(),
                     f"rmat{scale}(synthetic)")


def rmat_device_shard(scale: int, edge_factor: int, seed: int, rank: int, world: int, device,
                      abcd=(0.57, 0.19, 0.19, 0.05), chunk_log2: int = 24, row_weight: int = None) -> DeviceCsr:
    """Row block `rank` of ONE global 2^scale R-MAT cut into `world` cost-balanced row blocks (strong scaling,
    BASELINE config 3), generated PER SHARD: the edge list is produced in fixed seeded chunks (the same for every
    world size, so N = 1 and N = 8 see the same matrix), pass 1 only histograms the rows to find the cuts
    (sharding.partition_rows_by_cost semantics: non-zeros + row_weight * rows per block; row_weight = 0 is the plain
    nnz balance), pass 2 regenerates the chunks and keeps the rows of this rank.
    The full matrix is never materialised on a rank that owns 1/world of it."""
    import torch

    n = 1 << scale
    ne = n * edge_factor
    chunk = min(ne, 1 << chunk_log2)
    nchunks = (ne + chunk - 1) // chunk
    a, b, c, _ = abcd

    def gen_chunk(ci: int):
        cnt = min(chunk, ne - ci * chunk)
        g = torch.Generator(device=device).manual_seed(seed * 1000003 + 7919 * ci)
        rows = torch.zeros(cnt, dtype=torch.int32, device=device)
        cols = torch.zeros(cnt, dtype=torch.int32, device=device)
        for _lvl in range(scale):
            r = torch.rand(cnt, generator=g, device=device)
            rows = (rows << 1) | (r >= (a + b)).to(torch.int32)
            cols = (cols << 1) | (((r >= a) & (r < a + b)) | (r >= a + b + c)).to(torch.int32)
        return rows, cols

    lo, hi = 0, n
    if row_weight is None:
        from .sharding import ROW_WEIGHT
        row_weight = ROW_WEIGHT
    if world > 1:
        hist = torch.zeros(n, dtype=torch.int64, device=device)
        for ci in range(nchunks):
            rows, _ = gen_chunk(ci)
            hist += torch.bincount(rows, minlength=n)
        ptr = torch.zeros(n + 1, dtype=torch.int64, device=device)
        ptr[1:] = torch.cumsum(hist, 0)
        del hist
        ptr += row_weight * torch.arange(n + 1, dtype=torch.int64, device=device)  # cost up to row r
        total = ne + row_weight * n
        cuts = [0]
        for gidx in range(1, world):
            target = (gidx * total) // world
            r = int(torch.searchsorted(ptr, torch.tensor([target], device=device), right=True)[0]) - 1
            cuts.append(min(max(r, cuts[-1]), n))
        cuts.append(n)
        lo, hi = cuts[rank], cuts[rank + 1]
        del ptr
    keep_r, keep_c = [], []
    for ci in range(nchunks):
        rows, cols = gen_chunk(ci)
        if world > 1:
            sel = (rows >= lo) & (rows < hi)
            rows, cols = rows[sel] - lo, cols[sel]
        keep_r.append(rows)
        keep_c.append(cols)
    rows = torch.cat(keep_r)
    cols = torch.cat(keep_c)
    del keep_r, keep_c
    rows, order = torch.sort(rows, stable=True)  # file order inside a row is kept (the reference's counting sort)
    cols = cols[order].contiguous()
    del order
    mloc = hi - lo
    counts = torch.bincount(rows, minlength=mloc)
    del rows
    row_ptr = torch.zeros(mloc + 1, dtype=torch.int64, device=device)
    row_ptr[1:] = torch.cumsum(counts, 0)
    name = (f"rmat{scale}(synthetic)" if world == 1 else
            f"rmat{scale}(synthetic) rows {lo}:{hi} of {world} blocks balanced by nnz + {row_weight} * rows")
    return DeviceCsr(mloc, n, int(cols.numel()), row_ptr.to(torch.int32), cols, name)

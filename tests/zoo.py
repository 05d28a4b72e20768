"""Seeded matrix zoo shared by the CPU and GPU parity tests.

Covers the edge cases the reference handles: empty rows (incl. ~50 % empty,
leading and trailing), very long rows spanning many tiles, p = 1 (nnz < omega*sigma), nnz an exact
multiple of omega*sigma, non-square, a dense diagonal (every element starts a row), rows that begin
exactly on tile boundaries.
"""
import numpy as np

from benchmark_spmv_using_csr5_amd import matrices as M


def _lens(name, lengths, n, seed, band=0.0):
    rng = np.random.default_rng(seed)
    return M.csr_from_row_lengths(np.asarray(lengths), n, rng, band=band, name=name)


def kat0():
    # SURVEY.md section 8(a) worked known-answer: row lengths 3,0,5,1,9,0,0,14,2
    return _lens("kat0", [3, 0, 5, 1, 9, 0, 0, 14, 2], 9, 11)


def small_zoo():
    rng = np.random.default_rng(1234)
    z = [
        kat0(),
        M.example_matrix(),
        _lens("tiny-p1", [2, 0, 3, 1, 5, 0, 4, 4], 8, 1),                      # 19 nnz, p = 1
        _lens("dense16", [16] * 16, 16, 2),                                     # nnz = 256 = 64*4
        _lens("nonsquare", rng.integers(0, 12, size=200), 150, 3),
        _lens("half-empty", rng.integers(1, 7, size=6000) * (rng.random(6000) < 0.5), 5000, 4),
        _lens("hub", [0] * 5 + [9000] + [0] * 40 + [1, 2, 3] + [0] * 40, 4000, 5),
        _lens("diag", [1] * 5000, 5000, 6),
        _lens("lead-trail-empty", [0] * 300 + list(rng.integers(0, 9, size=3000)) + [0] * 700, 4000, 7),
        _lens("aligned64", [64] * 300, 512, 8),                                 # rows start on tile edges
        _lens("aligned1024", [1024] * 20 + [3, 0, 2], 2048, 9),
        _lens("two-hubs", [5000, 0, 0, 7000, 1, 1, 0, 0, 2], 8000, 10),
        _lens("one-row", [3000], 3000, 12),
        _lens("single-nnz", [0, 0, 1, 0], 4, 13),
        M.scircuit_like(scale=0.03),
        M.webbase_like(scale=0.01),
        M.rmat(10, 8, seed=21),
    ]
    nd = M.nd24k_like(scale=0.005, dtype=np.float64)
    z.append(nd)
    return z


def empty_matrix():
    return M.CsrMatrix(5, 5, np.zeros(6, dtype=np.int32), np.zeros(0, dtype=np.int32),
                       np.zeros(0, dtype=np.float64), "all-empty")

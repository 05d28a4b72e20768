#!/usr/bin/env python3
"""LDS y-segment compaction on / off by mean row length k (columns within +-64 of the diagonal, ~4 M non-zeros, rule sigma):
where does staging a 4-KB slice of x per tile pay?  us per SpMV, fused mode, hipGraph replay."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from benchmark_spmv_using_csr5_amd import handle as H, matrices as M  # noqa: E402

DEV = torch.device("cuda:0")


def run(mat, val, x, dtype, xwin):
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    rp, ci, va, xd = (torch.from_numpy(a).to(DEV) for a in (mat.row_ptr, mat.col, val, x))
    yd = torch.zeros(mat.m, dtype=tdt, device=DEV)
    A = H.anonymouslibHandle(mat.m, mat.n, dtype=np.dtype(dtype).name)
    A.inputCSR(mat.nnz, rp, ci, va); A.setX(xd); A.setSigma(-1); A.setLdsY(xwin); A.asCSR5()
    A.spmv_repeat(1.0, yd, 200)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        A.timer_start(); A.spmv_repeat(1.0, yd, 200); best = min(best, A.timer_stop() / 200 * 1e3)
    cover = A.info().x_window_cover_pct
    A.destroy(); A.close()
    return best, cover


rng = np.random.default_rng(3)
for band in (0.0, 1.0):
    for k in (2, 4, 8, 16, 24, 32, 48, 64, 128):
        m = max(4_000_000 // k, 2048)
        mat = M.csr_from_row_lengths(rng.poisson(k, size=m).astype(np.int64), m, rng, band=band, name=f"k{k}")
        for dtype in (np.float64, np.float32):
            val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=k, mode="int")
            off, cover = run(mat, val, x, dtype, 0)
            on, _ = run(mat, val, x, dtype, 2)
            print(f"band={band} k={k:4d} {np.dtype(dtype).name:7s} cover={cover:3d}%  off {off:7.2f}  on {on:7.2f}  on/off {on / off:5.2f}", flush=True)

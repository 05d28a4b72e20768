#!/usr/bin/env python3
"""sigma per row-length class on gfx950: for synthetic matrices of ~4 M non-zeros with mean
row length k, time every candidate sigma (fused mode, hipGraph replay) and compare the rule's pick
(csr5hip_auto_sigma) with the measured best.
usage: python scripts/experiments/sigma_table.py > profiles/rNN_sigma_table.txt"""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from benchmark_spmv_using_csr5_amd import _capi, handle as H, matrices as M  # noqa: E402

DEV = torch.device("cuda:0")
CANDIDATES = [4, 5, 6, 8, 10, 12, 16, 20, 24, 32]


def time_sigma(mat, val, x, sigma, dtype):
    tdt = torch.float64 if dtype == np.float64 else torch.float32
    rp = torch.from_numpy(mat.row_ptr).to(DEV)
    ci = torch.from_numpy(mat.col).to(DEV)
    va = torch.from_numpy(val).to(DEV)
    xd = torch.from_numpy(x).to(DEV)
    yd = torch.zeros(mat.m, dtype=tdt, device=DEV)
    A = H.anonymouslibHandle(mat.m, mat.n, dtype=np.dtype(dtype).name)
    assert A.inputCSR(mat.nnz, rp, ci, va) == 0 and A.setX(xd) == 0
    assert A.setSigma(sigma) == 0 and A.asCSR5() == 0
    reps = 200
    A.spmv_repeat(1.0, yd, reps)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        A.timer_start()
        A.spmv_repeat(1.0, yd, reps)
        best = min(best, A.timer_stop() / reps * 1e3)
    used = A.info().sigma
    A.destroy()
    A.close()
    return best, used


def main():
    lib = _capi.load()
    print("# mean row length k | dtype | columns | rule sigma: us | best sigma: us | rule is slower by")
    rng = np.random.default_rng(3)
    for k in (2, 3, 4, 6, 8, 12, 16, 24, 32, 64, 128, 256, 512):
        m = max(4_000_000 // k, 2048)
        for band, cname in ((0.0, "random"), (1.0, "near diagonal")):
            lens = rng.poisson(k, size=m).astype(np.int64)
            mat = M.csr_from_row_lengths(lens, m, rng, band=band, name=f"k{k}")
            for dtype in (np.float64, np.float32):
                val, x = M.fill_values(mat.nnz, mat.n, dtype, seed=k, mode="int")
                rule = lib.csr5hip_auto_sigma(mat.m, mat.nnz, _capi.F64 if dtype == np.float64 else _capi.F32)
                times = {}
                for s in CANDIDATES + ([rule] if rule not in CANDIDATES else []):
                    times[s], _ = time_sigma(mat, val, x, s, dtype)
                best = min(times, key=times.get)
                print(f"k={k:4d} | {np.dtype(dtype).name:7s} | {cname:13s} | rule {rule:2d}: {times[rule]:8.2f} | "
                      f"best {best:2d}: {times[best]:8.2f} | {100 * (times[rule] / times[best] - 1):5.1f} %   "
                      + " ".join(f"{s}:{times[s]:.1f}" for s in sorted(times)), flush=True)


if __name__ == "__main__":
    main()

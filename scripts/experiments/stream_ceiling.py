#!/usr/bin/env python3
"""Streaming ceiling of the tile kernels on an R-MAT-sized matrix whose gathers are free: the rows of R-MAT `scale`, every
column redrawn within +-32 of the row index (x lines stay in L1/L2).  Plain one-tile kernel (sigma 16 and 8) vs the
persistent hot-table kernel forced on the same matrix.

    python scripts/experiments/stream_ceiling.py --scale 22
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} -> {rc}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=22)
    args = ap.parse_args()
    import torch
    from benchmark_spmv_using_csr5_amd import handle as H
    from benchmark_spmv_using_csr5_amd import matrices as M

    dev = torch.device("cuda:0")
    mat = M.rmat_device_shard(args.scale, 16, 1, 0, 1, dev)
    g = torch.Generator(device=dev).manual_seed(5)
    rows = torch.repeat_interleave(torch.arange(mat.m, device=dev, dtype=torch.int64),
                                   (mat.row_ptr[1:] - mat.row_ptr[:-1]).to(torch.int64))
    col = ((rows + torch.randint(-32, 33, (mat.nnz,), generator=g, device=dev)) % mat.n).to(torch.int32)
    del rows
    b_alg = M.algorithmic_bytes(mat.m, mat.n, mat.nnz, 8)
    for label, sigma, slabs, hot, nt in (("plain kernel, sigma 16", 16, 0, 0, 1), ("plain kernel, sigma 8", 8, 0, 0, 1),
                                         ("plain kernel, sigma 16, no NT", 16, 0, 0, 0),
                                         ("hot-table kernel (forced), 8 slabs", 16, 8, 2, 1),
                                         ("slab child without table, 8 slabs", 16, 8, 0, 1)):
        va = torch.randint(0, 10, (mat.nnz,), generator=g, device=dev).to(torch.float64)
        x = torch.randint(0, 10, (mat.n,), generator=g, device=dev).to(torch.float64)
        y = torch.zeros(mat.m, dtype=torch.float64, device=dev)
        A = H.anonymouslibHandle(mat.m, mat.n)
        ck(A.inputCSR(mat.nnz, mat.row_ptr, col.clone(), va), "inputCSR")
        ck(A.setX(x), "setX")
        A.setSigma(sigma)
        A.setColumnSlabs(slabs)
        A.setSlabHot(hot)
        A.setStreamNT(2 if nt else 0)
        ck(A.asCSR5(), "asCSR5")
        i = A.info()
        ck(A.spmv_repeat(1.0, y, 10), "spmv_repeat")
        ck(A.spmv_repeat(1.0, y, 50), "spmv_repeat")  # instantiates the timed call's graph outside the timed region
        torch.cuda.synchronize()
        A.timer_start()
        ck(A.spmv_repeat(1.0, y, 50), "spmv_repeat")
        us = A.timer_stop() * 1e3 / 50
        print(json.dumps({"kernel": label, "nnz": mat.nnz, "slabs": i.column_slabs, "hot": i.slab_hot,
                          "cover_pct": i.slab_hot_cover_pct, "us": round(us, 1),
                          "B_alg_TBps": round(b_alg / us / 1e6, 2)}))
        A.destroy()
        A.close()


if __name__ == "__main__":
    main()

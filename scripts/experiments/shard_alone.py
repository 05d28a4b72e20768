#!/usr/bin/env python3
"""One row block of the strong-scaling R-MAT (rank r of W) ALONE on one GPU: what each GPU of an N-GPU run does per
SpMV, measured without the other ranks (a 1-GPU box cannot run them side by side).  Ideal = t(1 of 1) / W.

    python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7
"""
import argparse
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} -> {rc}")


def measure_block(scale, world, rank, dev, steps=50, slabs="auto", hot="auto", row_weight=None, x_snapshot=0, seed=1,
                  value_seed=7, keep_y=False):
    """Row block `rank` of the `world`-way cost-balanced cut of ONE R-MAT `scale` (the strong-scaling shard of BASELINE config 4)
    through its own handle at the library's defaults; returns (record, y or None).  x and the values are drawn from `value_seed`
    over the GLOBAL index space (the same x for every block; a block's values = its slice of one global value vector would need
    the global non-zero offset, so values are seeded per block: value_seed + rank)."""
    import torch
    from benchmark_spmv_using_csr5_amd import handle as H
    from benchmark_spmv_using_csr5_amd import matrices as M

    mat = M.rmat_device_shard(scale, 16, seed, rank, world, dev, row_weight=row_weight)
    gx = torch.Generator(device=dev).manual_seed(value_seed)
    x = torch.randint(0, 10, (mat.n,), generator=gx, device=dev).to(torch.float64)
    gv = torch.Generator(device=dev).manual_seed(value_seed + 1 + rank)
    va = torch.randint(0, 10, (mat.nnz,), generator=gv, device=dev).to(torch.float64)
    y = torch.zeros(mat.m, dtype=torch.float64, device=dev)
    A = H.anonymouslibHandle(mat.m, mat.n)
    ck(A.inputCSR(mat.nnz, mat.row_ptr, mat.col.clone(), va), "inputCSR")
    ck(A.setX(x), "setX")
    A.setSigma(-1)
    A.setColumnSlabs(1 if slabs == "auto" else int(slabs))
    A.setSlabHot({"off": 0, "auto": 1, "force": 2}[hot])
    A.setXSnapshot(x_snapshot)
    ck(A.asCSR5(), "asCSR5")
    i = A.info()
    ck(A.spmv_repeat(1.0, y, 10), "spmv_repeat")
    # the graph of the timed call (one graph per count) is captured and instantiated HERE, not inside the timed region
    # (until round 4's last day it was: +10-12 us per step on these 170-us steps)
    ck(A.spmv_repeat(1.0, y, steps), "spmv_repeat")
    torch.cuda.synchronize()
    A.timer_start()
    ck(A.spmv_repeat(1.0, y, steps), "spmv_repeat")
    us = A.timer_stop() * 1e3 / steps
    b_alg = M.algorithmic_bytes(mat.m, mat.n, mat.nnz, 8)
    rec = {"rank": rank, "world": world, "m": mat.m, "nnz": mat.nnz, "sigma": i.sigma, "slabs": i.column_slabs,
           "hot": i.slab_hot, "hot_cover_pct": i.slab_hot_cover_pct, "x_snapshot": x_snapshot, "us": round(us, 1),
           "frac": round(b_alg / (us * 1e-6) / 8e12, 3),
           "gflops_if_all_ranks_alike": round(2 * mat.nnz * world / (us * 1e-6) / 1e9, 1)}
    A.destroy()  # (asCSR: the value array the handle transposed in place is back in CSR order)
    A.close()
    out = None
    if keep_y:
        torch.cuda.synchronize()
        out = {"y": y.cpu().numpy(), "row_ptr": mat.row_ptr.cpu().numpy(), "col": mat.col.cpu().numpy(),
               "val": va.cpu().numpy(), "x": x.cpu().numpy()}
    del mat, va, x, y
    torch.cuda.empty_cache()
    return rec, out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=24)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--ranks", default="0")
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--slabs", default="auto")
    ap.add_argument("--hot", default="auto")
    ap.add_argument("--row-weight", type=int, default=None)
    ap.add_argument("--x-snapshot", type=int, default=0, help="0 = the library default (x read live); 1 = as bench.py --x-snapshot 1")
    args = ap.parse_args()
    import torch

    dev = torch.device("cuda:0")
    for rank in [int(r) for r in args.ranks.split(",")]:
        rec, _ = measure_block(args.scale, args.world, rank, dev, args.steps, args.slabs, args.hot, args.row_weight,
                               args.x_snapshot)
        print(json.dumps(rec))


if __name__ == "__main__":
    main()

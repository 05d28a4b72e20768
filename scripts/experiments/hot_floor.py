#!/usr/bin/env python3
"""What the persistent hot kernel costs WITHOUT cold gathers: the rows of R-MAT `scale`, columns redrawn from `hubs`
popular columns (all of them fit the tables: coverage 100 %), against the same rows with their real columns.

    python scripts/experiments/hot_floor.py --scale 22 --hubs 65536
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
    ap.add_argument("--hubs", type=int, default=65536)
    ap.add_argument("--slabs", default="auto")
    args = ap.parse_args()
    import torch
    from benchmark_spmv_using_csr5_amd import handle as H
    from benchmark_spmv_using_csr5_amd import matrices as M

    dev = torch.device("cuda:0")
    mat = M.rmat_device_shard(args.scale, 16, 1, 0, 1, dev)
    g = torch.Generator(device=dev).manual_seed(5)
    hub_ids = torch.randperm(mat.n, generator=g, device=dev)[: args.hubs].to(torch.int32)
    for kind in ("real columns", f"{args.hubs} hub columns only", "90% hub columns"):
        if kind == "real columns":
            col = mat.col.clone()
        else:
            col = hub_ids[torch.randint(0, args.hubs, (mat.nnz,), generator=g, device=dev)]
            if kind.startswith("90%"):
                keep = torch.rand(mat.nnz, generator=g, device=dev) < 0.1
                col = torch.where(keep, mat.col, col)
        va = torch.randint(0, 10, (mat.nnz,), generator=g, device=dev).to(torch.float64)
        x = torch.randint(0, 10, (mat.n,), generator=g, device=dev).to(torch.float64)
        y = torch.zeros(mat.m, dtype=torch.float64, device=dev)
        A = H.anonymouslibHandle(mat.m, mat.n)
        ck(A.inputCSR(mat.nnz, mat.row_ptr, col, va), "inputCSR")
        ck(A.setX(x), "setX")
        A.setSigma(-1)
        A.setColumnSlabs(1 if args.slabs == "auto" else int(args.slabs))
        A.setSlabHot(2)
        ck(A.asCSR5(), "asCSR5")
        i = A.info()
        ck(A.spmv_repeat(1.0, y, 10), "spmv_repeat")
        ck(A.spmv_repeat(1.0, y, 50), "spmv_repeat")  # instantiates the timed call's graph outside the timed region
        torch.cuda.synchronize()
        A.timer_start()
        ck(A.spmv_repeat(1.0, y, 50), "spmv_repeat")
        us = A.timer_stop() * 1e3 / 50
        print(json.dumps({"columns": kind, "nnz": mat.nnz, "slabs": i.column_slabs, "hot": i.slab_hot,
                          "cover_pct": i.slab_hot_cover_pct, "segments": i.slab_segments, "us": round(us, 1)}))
        A.destroy()
        A.close()


if __name__ == "__main__":
    main()

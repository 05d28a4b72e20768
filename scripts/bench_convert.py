#!/usr/bin/env python3
"""CSR -> CSR5 conversion time in the reference CLI's protocol (CSR5_avx2/main.cpp:41-52: convert back and forth a few
times, then time asCSR5), repeated: median / min / max of `--rounds` timed asCSR5 calls, each followed by an untimed
asCSR.  asCSR5 returns after its single stream synchronisation, so the host clock around the call is the whole cost.

    python scripts/bench_convert.py --workload scircuit --rounds 50
"""
import argparse
import json
import os
import statistics
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def ck(rc, what):
    if rc != 0:
        raise RuntimeError(f"{what} -> {rc}")


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="scircuit", choices=["scircuit", "webbase", "nd24k", "rmat20", "rmat22", "rmat24"])
    ap.add_argument("--rounds", type=int, default=50)
    ap.add_argument("--sigma", type=int, default=-1)
    ap.add_argument("--slabs", default="auto")
    args = ap.parse_args()
    import torch
    from benchmark_spmv_using_csr5_amd import handle as H
    from benchmark_spmv_using_csr5_amd import matrices as M

    dev = torch.device("cuda:0")
    f32 = args.workload == "nd24k"
    if args.workload.startswith("rmat"):
        mat = M.rmat_device_shard(int(args.workload[4:]), 16, 1, 0, 1, dev)
        rp, ci = mat.row_ptr, mat.col.clone()
    else:
        mat = {"scircuit": M.scircuit_like, "webbase": M.webbase_like, "nd24k": M.nd24k_like}[args.workload]()
        rp, ci = torch.from_numpy(mat.row_ptr).to(dev), torch.from_numpy(mat.col).to(dev)
    dt = torch.float32 if f32 else torch.float64
    va = torch.randint(0, 10, (mat.nnz,), device=dev).to(dt)
    x = torch.randint(0, 10, (mat.n,), device=dev).to(dt)
    A = H.anonymouslibHandle(mat.m, mat.n, dtype="float32" if f32 else "float64")
    ck(A.inputCSR(mat.nnz, rp, ci, va), "inputCSR")
    ck(A.setX(x), "setX")
    A.setSigma(args.sigma)
    A.setColumnSlabs(1 if args.slabs == "auto" else int(args.slabs))
    A.warmup()
    for _ in range(5):
        ck(A.asCSR5(), "asCSR5")
        ck(A.asCSR(), "asCSR")
    torch.cuda.synchronize()
    times = []
    for _ in range(args.rounds):
        t0 = time.perf_counter()
        rc = A.asCSR5()
        t1 = time.perf_counter()
        ck(rc, "asCSR5")
        times.append((t1 - t0) * 1e6)
        info = A.info()
        ck(A.asCSR(), "asCSR")
    out = {"workload": args.workload, "m": mat.m, "nnz": mat.nnz, "sigma": info.sigma, "tiles": info.p,
           "column_slabs": info.column_slabs, "rounds": args.rounds,
           "as_csr5_us": {"median": round(statistics.median(times), 1), "min": round(min(times), 1),
                          "max": round(max(times), 1)}}
    print(json.dumps(out))
    A.close()


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Coupled iterations: power iteration x <- A x / |A x| on a square R-MAT cut into
cost-balanced (nnz + 2 per row) row blocks, one rank per GPU.  Per iteration: CSR5 SpMV of the local block written straight into
this rank's slot of the next x, ONE in-place all-gather over RCCL/xGMI, a dot and a norm.

  python scripts/bench_coupled.py --scale 20 --iters 200
  python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 scripts/bench_coupled.py --scale 22
Prints one JSON line (rank 0).  CSR5_BENCH_SHARE_GPU=1 puts every rank on cuda:0 with gloo (1-GPU box check).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--edge-factor", type=int, default=16)
    ap.add_argument("--iters", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--graph", action="store_true", help="replay the iteration from one hipGraph (pairs of steps)")
    args = ap.parse_args()

    import torch
    from benchmark_spmv_using_csr5_amd import matrices as M, sharding as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    share = os.environ.get("CSR5_BENCH_SHARE_GPU") == "1"
    dev = torch.device("cuda", 0 if share else local)
    torch.cuda.set_device(dev)
    dist = None
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("gloo" if share else "nccl", rank=rank, world_size=world)

    mat = M.rmat(scale=args.scale, edge_factor=args.edge_factor, seed=args.seed)
    val, x0 = M.fill_values(mat.nnz, mat.n, np.float64, seed=args.seed + 1, mode="pos")
    x0 = x0 / np.linalg.norm(x0)
    cp = S.CoupledSpmv(mat.row_ptr, mat.col, val, mat.n, rank, world)
    run = S.hip_coupled_spmv(dev)
    a = torch.from_numpy(cp.layout.to_padded(x0)).to(dev)
    b = torch.zeros_like(a)

    if args.graph:
        # capture happens inside; time a second call's replays only by timing a long run minus nothing: the capture
        # cost is amortised over --iters (reported separately below)
        t_cap = time.perf_counter()
        cp.power_iteration_graph(run, a, b, 2)
        torch.cuda.synchronize()
        t_cap = time.perf_counter() - t_cap
    else:
        cp.power_iteration(run, a, b, args.warmup)
    a.copy_(torch.from_numpy(cp.layout.to_padded(x0)).to(dev))
    b.zero_()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    if args.graph:
        xk, lam = cp.power_iteration_graph(S.hip_coupled_spmv(dev), a, b, args.iters)
    else:
        xk, lam = cp.power_iteration(run, a, b, args.iters)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=dev if not share else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    if rank == 0:
        print(json.dumps({
            "metric": "coupled power iteration", "unit": "iterations/s", "value": round(args.iters / dt, 1),
            "n_gpus": world, "us_per_iteration": round(dt / args.iters * 1e6, 2),
            "gflops": round(2.0 * mat.nnz * args.iters / dt * 1e-9, 1),
            "config": {"workload": f"R-MAT scale {args.scale} ef {args.edge_factor} (synthetic), row blocks by nnz",
                       "m": mat.m, "nnz": mat.nnz, "slot_width": cp.layout.width,
                       "allgather_bytes_per_iteration": cp.layout.padded_len * 8, "backend": "gloo(shared GPU)" if share else "rccl"},
            "rayleigh": float(lam), "scaling": "strong",
            "launch": "one hipGraph per two iterations (capture included in the time)" if args.graph else "eager"}), flush=True)
    run.state["A"].destroy()
    run.state["A"].close()
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()

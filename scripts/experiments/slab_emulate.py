#!/usr/bin/env python3
"""Experiment (round 2): does a column-slab stacked matrix cut the x-gather over-fetch?

Emulates the planned kernel-side structure with torch as plumbing: the non-zeros are stably partitioned by
slab(col) = xor-fold of (col >> shift) into S slabs, rows are compacted per slab, and the S sub-matrices are
stacked vertically into ONE CSR matrix A' (m' = number of (row, slab) segments).  A' is converted and run by the
existing handle (XCD-contiguous tile ranges ~ one slab range per XCD).  Reports the kernel time of A'x and the
segment count (the combine step y[r] = sum of its <= S partials is modelled, not run, here).

    python scripts/experiments/slab_emulate.py --workload rmat22 --slabs 8,16 --shifts 4,9 --sigmas -1,8
"""
import argparse
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from benchmark_spmv_using_csr5_amd import handle as H  # noqa: E402
from benchmark_spmv_using_csr5_amd import matrices as M  # noqa: E402


def xor_fold(v, bits):
    out = torch.zeros_like(v)
    mask = (1 << bits) - 1
    while True:
        out ^= v & mask
        v = v >> bits
        if not bool((v != 0).any()):
            break
    return out


def time_handle(m, n, nnz, rp, ci, va, xd, sigma, steps, nt="auto"):
    yd = torch.zeros(m, dtype=va.dtype, device=va.device)
    A = H.anonymouslibHandle(m, n, dtype="float64" if va.dtype == torch.float64 else "float32")
    assert A.inputCSR(nnz, rp, ci, va) == 0
    assert A.setX(xd) == 0
    assert A.setSigma(sigma) == 0
    assert A.setStreamNT({"off": 0, "auto": 1, "force": 2}[nt]) == 0
    assert A.asCSR5() == 0
    info = A.info()
    assert A.spmv(1.0, yd) == 0
    torch.cuda.synchronize()
    y = yd.clone()
    assert A.spmv_repeat(1.0, yd, steps) == 0
    torch.cuda.synchronize()
    A.timer_start()
    assert A.spmv_repeat(1.0, yd, steps) == 0
    ms = A.timer_stop() / steps
    A.destroy()
    A.close()
    return ms, info.sigma, info.p, y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="rmat22")
    ap.add_argument("--slabs", default="8,16")
    ap.add_argument("--shifts", default="4,9")
    ap.add_argument("--sigmas", default="-1")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--nt", default="auto")
    ap.add_argument("--baseline", type=int, default=1)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    if args.workload.startswith("rmat"):
        mat = M.rmat_device(int(args.workload[4:]), 16, 1, 0, 1, dev)
        rp, ci = mat.row_ptr, mat.col
        m, n, nnz = mat.m, mat.n, mat.nnz
    else:
        gen = {"scircuit": M.scircuit_like, "webbase": M.webbase_like}[args.workload]
        hm = gen(dtype=np.float64)
        m, n, nnz = hm.m, hm.n, hm.nnz
        rp = torch.from_numpy(hm.row_ptr).to(dev)
        ci = torch.from_numpy(hm.col).to(dev)
    g = torch.Generator(device=dev).manual_seed(5)
    va = torch.randint(0, 10, (nnz,), generator=g, device=dev).to(torch.float64)
    xd = torch.randint(0, 10, (n,), generator=g, device=dev).to(torch.float64)
    b_alg = M.algorithmic_bytes(m, n, nnz, 8)
    print(f"{args.workload}: m={m} n={n} nnz={nnz} B_alg={b_alg/1e6:.1f} MB", flush=True)
    y_ref = None
    if args.baseline:
        for s in [int(v) for v in args.sigmas.split(",")]:
            ms, sg, p, y_ref = time_handle(m, n, nnz, rp, ci.clone(), va.clone(), xd, s, args.steps, args.nt)
            print(f"  baseline sigma={sg} tiles={p}: {ms*1e3:.1f} us  frac={b_alg/ms/1e6/8000:.3f}", flush=True)
    rows = torch.repeat_interleave(torch.arange(m, device=dev, dtype=torch.int32),
                                   (rp[1:] - rp[:-1]).to(torch.int64))
    for S in [int(v) for v in args.slabs.split(",")]:
        bits = S.bit_length() - 1
        for shift in [int(v) for v in args.shifts.split(",")]:
            slab = xor_fold((ci >> shift).to(torch.int32), bits).to(torch.int16)
            counts = torch.bincount(slab.to(torch.int64), minlength=S)
            slab_s, perm = torch.sort(slab, stable=True)
            del slab
            rows_p = rows[perm]
            ci_p = ci[perm].contiguous()
            va_p = va[perm].contiguous()
            del perm
            start = torch.ones(nnz, dtype=torch.bool, device=dev)
            start[1:] = (rows_p[1:] != rows_p[:-1]) | (slab_s[1:] != slab_s[:-1])
            del slab_s
            pos = torch.nonzero(start).flatten()
            mp = int(pos.numel())
            rp_p = torch.empty(mp + 1, dtype=torch.int32, device=dev)
            rp_p[:mp] = pos.to(torch.int32)
            rp_p[mp] = nnz
            seg_row = rows_p[pos].to(torch.int64)
            del start, pos, rows_p
            for s in [int(v) for v in args.sigmas.split(",")]:
                ms, sg, p, P = time_handle(mp, n, nnz, rp_p, ci_p.clone(), va_p.clone(), xd, s, args.steps, args.nt)
                ok = ""
                if y_ref is not None:
                    yy = torch.zeros(m, dtype=torch.float64, device=dev)
                    yy.index_add_(0, seg_row, P)
                    ok = " exact" if bool((yy == y_ref).all()) else " MISMATCH"
                # modelled combine: write P (in the main kernel, counted there), read P + masks, write y
                comb_bytes = 8 * mp + 1.5 * m + 8 * m
                comb_ms = comb_bytes / 5.0e9  # at 5 TB/s
                tot = ms + comb_ms
                print(f"  S={S} shift={shift} sigma={sg} tiles={p} segments={mp} ({mp/nnz:.3f}/nnz) "
                      f"imbalance={float(counts.max())/float(counts.float().mean()):.3f}: "
                      f"main {ms*1e3:.1f} us + combine~{comb_ms*1e3:.1f} us = {tot*1e3:.1f} us  "
                      f"frac={b_alg/tot/1e6/8000:.3f}{ok}", flush=True)
            del ci_p, va_p, rp_p, seg_row


if __name__ == "__main__":
    main()

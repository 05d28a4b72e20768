"""Same-process A/B of the plain-path kernels on the SuiteSparse-shaped stand-ins: one tile per wavefront (k_spmv) vs the
range-walking pipelined kernel (k_spmv_walk) at several range counts; warm (back-to-back on one matrix) and cold (rotating
copies beyond the Infinity Cache) HIP-event microseconds per SpMV.  Usage: python walk_ab.py [workload ...] [--ranges a,b,c]"""
import argparse
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench as B  # noqa: E402
from benchmark_spmv_using_csr5_amd import matrices as M  # noqa: E402


def base_args(**kw):
    a = types.SimpleNamespace(values="int", sigma="-1", mode="fused", x_window="auto", xcd_remap=1, lds_y="auto", stream_nt="auto",
                              slabs="auto", slab_shift=None, slab_hot="auto", x_snapshot=0, zero_empty=0, tile_walk="auto",
                              walk_ranges=0, seed=1)
    a.__dict__.update(kw)
    return a


def measure(mat, label, dtype_name, a, dev, steps=400, cold=True):
    prob = B.Problem(mat, label, dtype_name, a, dev, 14)
    wall, ev = B.timed(prob, steps, 50, "graph")
    warm_us = ev / steps * 1e3
    cold_us = None
    if cold:
        cms, k, cs = B.timed_cold(lambda: B.Problem(mat, label, dtype_name, a, dev, 14), prob, steps, 20)
        cold_us = cms * 1e3
    i = prob.info
    desc = f"sigma={i.sigma} walk={i.tile_walk}/{i.walk_ranges} c16={i.narrow_columns} xwin={i.x_window_active}/{i.walk_x_window}({i.walk_x_window_cover_pct}%) slabs={i.column_slabs}/hot={i.slab_hot} p={i.p}"
    b = prob.b_alg
    prob.close()
    return warm_us, cold_us, desc, b


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["nd24k", "scircuit", "webbase"])
    ap.add_argument("--ranges", default="1024,2048,3072,4096,8192")
    ap.add_argument("--no-cold", action="store_true")
    ap.add_argument("--sigma", default="-1")
    ap.add_argument("--slabs", default="auto")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for w in args.workloads:
        dtype_name = "f32" if w in ("nd24k", "nd24kx4") else "f64"
        npd = np.float32 if dtype_name == "f32" else np.float64
        mat = {"nd24k": lambda: M.nd24k_like(dtype=npd), "nd24kx4": lambda: M.nd24k_like(scale=4.0, dtype=npd), "scircuit": lambda: M.scircuit_like(dtype=npd),
               "webbase": lambda: M.webbase_like(dtype=npd),
               "rmat20": lambda: M.rmat(20, 16, seed=4, dtype=npd)}[w]()
        rows = []
        variants = [("one-tile", dict(tile_walk="off"))]
        for r in args.ranges.split(","):
            variants.append((f"walk/{r}", dict(tile_walk="force", walk_ranges=int(r))))
        for name, kw in variants:
            a = base_args(sigma=args.sigma, slabs=args.slabs, **kw)
            warm, cold, desc, b = measure(mat, w, dtype_name, a, dev, cold=not args.no_cold)
            frac_w = b / (warm * 1e-6) / 8e12
            frac_c = b / (cold * 1e-6) / 8e12 if cold else float("nan")
            print(f"{w:9s} {name:11s} warm {warm:8.2f} us ({frac_w:.3f})  cold {cold if cold else float('nan'):8.2f} us ({frac_c:.3f})  {desc}",
                  flush=True)


if __name__ == "__main__":
    main()

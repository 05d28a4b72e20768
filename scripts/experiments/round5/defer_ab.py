"""A/B of deferred carries (CSR5HIP_OPT_DEFER_CARRIES off / force / auto) on the stand-ins: warm and cold microseconds per
SpMV, plus a bit-for-bit comparison of the two results.  Usage: python defer_ab.py [workload ...] [--sigma s]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "round6"))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from locality import B, M, base_args  # noqa: E402  (the walking kernel's walk_ab.py, which this used to import, left the product)
from locality import measure as _measure  # noqa: E402


def measure(mat, label, dtype_name, a, dev, steps=400, cold=True):
    return _measure(mat, label, a, dev, steps, cold, dtype_name)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workloads", nargs="*", default=["nd24k", "scircuit", "webbase"])
    ap.add_argument("--sigma", default="-1")
    ap.add_argument("--slabs", default="auto")
    ap.add_argument("--no-cold", action="store_true")
    ap.add_argument("--modes", default="off,force,auto")
    ap.add_argument("--x-window", default="auto")
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    for w in args.workloads:
        dtype_name = "f32" if w in ("nd24k", "nd24kx4") else "f64"
        npd = np.float32 if dtype_name == "f32" else np.float64

        def fem(rows, per_row):
            rng = np.random.default_rng(5)
            return M.csr_from_row_lengths(np.full(rows, per_row), rows, rng, band=0.9, dtype=npd)

        slabs = "0" if w.endswith("p") else args.slabs

        if ":" in w:  # "nd24k:0.25", "nd24k64:0.25", "fem81:100000" -- the stand-in at another size
            name, arg = w.split(":")
            dtype_name = "f32" if name == "nd24k" else "f64"
            npd = np.float32 if dtype_name == "f32" else np.float64
            sized = {"nd24k": lambda: M.nd24k_like(scale=float(arg), dtype=npd), "nd24k64": lambda: M.nd24k_like(scale=float(arg), dtype=npd),
                     "fem27": lambda: fem(int(arg), 27), "fem81": lambda: fem(int(arg), 81), "fem200": lambda: fem(int(arg), 200),
                     "scircuit": lambda: M.scircuit_like(scale=float(arg), dtype=npd)}
        mat = {"nd24k": lambda: M.nd24k_like(dtype=npd), "nd24k64": lambda: M.nd24k_like(dtype=npd),
               "nd24kx4": lambda: M.nd24k_like(scale=4.0, dtype=npd), "fem27": lambda: fem(2_000_000, 27),
               "fem81": lambda: fem(1_000_000, 81), "fem200": lambda: fem(300_000, 200),
               "rmat22p": lambda: M.rmat(22, 16, seed=4, dtype=npd), "rmat20p": lambda: M.rmat(20, 16, seed=4, dtype=npd),
               "rmat18p": lambda: M.rmat(18, 16, seed=4, dtype=npd), "rmat19p": lambda: M.rmat(19, 16, seed=4, dtype=npd),
               "scircuit": lambda: M.scircuit_like(dtype=npd), "webbase": lambda: M.webbase_like(dtype=npd),
               "longrand": lambda: M.csr_from_row_lengths(np.random.default_rng(7).integers(500, 4000, size=6000), 400_000,
                                                          np.random.default_rng(8), band=0.0, dtype=npd),
               "mixed": lambda: M.csr_from_row_lengths(np.where(np.random.default_rng(9).random(400_000) < 0.02, 900, 12), 400_000,
                                                       np.random.default_rng(10), band=0.5, dtype=npd),
               "rmat20": lambda: M.rmat(20, 16, seed=4, dtype=npd), "rmat22": lambda: M.rmat(22, 16, seed=4, dtype=npd)}
        mat = (sized[w.split(":")[0]] if ":" in w else mat[w])()
        ys = {}
        for mode in args.modes.split(","):
            a = base_args(sigma=args.sigma, slabs=slabs, defer_carries=mode, x_window=args.x_window)
            warm, cold, desc, b = measure(mat, w, dtype_name, a, dev, cold=not args.no_cold)
            a.values = "real"  # (rounding-sensitive data for the comparison)
            prob = B.Problem(mat, w, dtype_name, a, dev, 14)
            prob.yd.fill_(float("nan"))
            prob.A.spmv(1.0, prob.yd)
            prob.A.spmv(1.0, prob.yd)
            torch.cuda.synchronize()
            ys[mode] = prob.yd.clone()
            deferred = prob.info.carries_deferred
            prob.close()
            fw = b / (warm * 1e-6) / 8e12
            fc = b / (cold * 1e-6) / 8e12 if cold else float("nan")
            print(f"{w:9s} defer={mode:5s} ({deferred}) warm {warm:8.2f} us ({fw:.3f})  cold {cold if cold else float('nan'):8.2f} us ({fc:.3f})  {desc}",
                  flush=True)
        if "off" in ys and "force" in ys:
            same = torch.equal(ys["off"].view(torch.uint8), ys["force"].view(torch.uint8))
            print(f"{w:9s} off == force bit for bit: {same}", flush=True)


if __name__ == "__main__":
    main()

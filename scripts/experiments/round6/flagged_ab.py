"""Same-call A/B of CSR5HIP_OPT_FLAGGED_COLUMNS (column words with the row-start flag in bit 31, no descriptor load) on the
short-row stand-ins: cold / warm microseconds with the option off and forced, several repetitions (the effect is small), plus a
bit-for-bit comparison of the results on real data.  Usage: python flagged_ab.py [--reps 3]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from locality import B, M, base_args, measure  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reps", type=int, default=3)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    cases = [("scircuit", M.scircuit_like(), {}), ("webbase/noslab", M.webbase_like(), dict(slabs="0")),
             ("webbase b0.9pl", M.webbase_like(band=0.9, far="powerlaw"), {}), ("scircuit b0.95", M.scircuit_like(band=0.95), {}),
             ("fem27 2M rows", M.csr_from_row_lengths(np.full(2_000_000, 27), 2_000_000, np.random.default_rng(5), band=0.9), dict(sigma="8"))]
    for name, mat, kw in cases:
        for rep in range(args.reps):
            row = []
            for mode in ("off", "force"):
                a = base_args(flagged_columns=mode, **kw)
                warm, cold, desc, b = measure(mat, name, a, dev, 400)
                row.append((mode, warm, cold))
            (_, w0, c0), (_, w1, c1) = row
            print(f"{name:16s} rep {rep}: off cold {c0:7.2f} warm {w0:7.2f} | flagged cold {c1:7.2f} warm {w1:7.2f} | "
                  f"cold {100 * (c1 / c0 - 1):+5.1f} % warm {100 * (w1 / w0 - 1):+5.1f} %  ({desc})", flush=True)
        ys = []
        for mode in ("off", "force"):
            a = base_args(flagged_columns=mode, values="real", **kw)
            prob = B.Problem(mat, name, "f64", a, dev, 14)
            prob.yd.fill_(float("nan"))
            prob.A.spmv(1.0, prob.yd)
            torch.cuda.synchronize()
            ys.append((prob.yd.clone(), prob.info.flagged_columns))
            prob.close()
        print(f"{name:16s} flagged_columns off/on = {ys[0][1]}/{ys[1][1]}; results bit-identical: "
              f"{torch.equal(ys[0][0].view(torch.uint8), ys[1][0].view(torch.uint8))}", flush=True)


if __name__ == "__main__":
    main()

"""The power-law BASELINE stand-ins on a LOCALITY axis: webbase-like at band 0.3 (the bench stand-in) / 0.6 /
0.9 and with power-law far columns, scircuit-like at 0.5 (bench) / 0.8 / 0.95 (+ power-law); per point the library's auto rules and
every forced path (plain, x-window, slabs without table, slabs + hot table), cold (rotating copies) and warm HIP-event
microseconds per SpMV.  Prints a markdown table and the bracket; --json writes the raw points.  A "mis-pick" is a point where the
auto rules' time is more than 5 % above the best forced path's (same call, same box).
Usage: python locality.py [--scale s] [--json out.json] [--md out.md] [--steps k] [--points webbase,scircuit]"""
import argparse
import json
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench as B  # noqa: E402
from benchmark_spmv_using_csr5_amd import matrices as M  # noqa: E402

POINTS = {
    "webbase": [(0.3, "uniform"), (0.6, "uniform"), (0.9, "uniform"), (0.3, "powerlaw"), (0.6, "powerlaw"), (0.9, "powerlaw")],
    "scircuit": [(0.5, "uniform"), (0.8, "uniform"), (0.95, "uniform"), (0.5, "powerlaw"), (0.8, "powerlaw")],
    # where the hot-table path starts to pay (skewed columns, growing size)
    "rmat19": [(0, "rmat")], "rmat20": [(0, "rmat")], "rmat21": [(0, "rmat")],
    "webbaseX2": [(0.3, "powerlaw")], "webbaseX4": [(0.3, "powerlaw"), (0.3, "uniform")], "webbaseX8": [(0.3, "powerlaw")],
}
# forced paths: name -> Problem args
PATHS = [
    ("auto", dict()),
    ("plain", dict(slabs="0", x_window="off")),
    ("x-window", dict(slabs="0", x_window="force")),
    ("slabs", dict(slabs="4", slab_hot="off")),
    ("slabs+table", dict(slabs="8", slab_hot="force")),
]


def base_args(**kw):
    a = types.SimpleNamespace(values="int", sigma="-1", mode="fused", x_window="auto", xcd_remap=1, lds_y="auto", stream_nt="auto",
                              slabs="auto", slab_shift=None, slab_hot="auto", x_snapshot=0, zero_empty=0, defer_carries="auto", seed=1)
    a.__dict__.update(kw)
    return a


def measure(mat, label, a, dev, steps, cold=True, dtype_name="f64"):
    prob = B.Problem(mat, label, dtype_name, a, dev, 14)
    _, ev = B.timed(prob, steps, 30, "graph")
    warm_us = ev / steps * 1e3
    cold_us = None
    if cold:
        cms, _, _ = B.timed_cold(lambda: B.Problem(mat, label, dtype_name, a, dev, 14), prob, steps, 20)
        cold_us = cms * 1e3
    i = prob.info
    desc = (f"sigma={i.sigma} xwin={int(i.x_window_active)}({i.x_window_cover_pct}%) slabs={i.column_slabs}"
            f"/hot={int(i.slab_hot)}({i.slab_hot_cover_pct}%) defer={int(i.carries_deferred)}")
    b = prob.b_alg
    prob.close()
    return warm_us, cold_us, desc, b


def make_matrix(kind, band, far, scale):
    """webbase / scircuit stand-ins on the locality axis; "rmatNN" = R-MAT scale NN (band / far ignored); "webbaseXk" = the webbase
    stand-in at k times its size (table-threshold points: where does the hot-table path start to pay?)"""
    if kind.startswith("rmat"):
        return M.rmat(int(kind[4:]), 16, seed=4)
    if kind.startswith("webbaseX"):
        return M.webbase_like(band=band, far=far, scale=scale * float(kind[8:]))
    gen = M.webbase_like if kind == "webbase" else M.scircuit_like
    return gen(band=band, far=far, scale=scale)


def run_point(kind, band, far, scale, dev, steps, paths=PATHS, cold=True):
    mat = make_matrix(kind, band, far, scale)
    out = {"kind": kind, "band": band, "far": far, "m": mat.m, "nnz": mat.nnz, "paths": {}}
    for name, kw in paths:
        try:
            warm, cold_us, desc, b = measure(mat, f"{kind}{band}{far}", base_args(**kw), dev, steps, cold=cold)
        except RuntimeError as e:  # a forced path the matrix cannot take (e.g. the table on a tiny x)
            out["paths"][name] = {"error": str(e)[:80]}
            continue
        if cold_us is None:  # (warm only: the decision test of tests/test_gpu_locality.py)
            cold_us = warm
        out["b_alg"] = b
        out["paths"][name] = {"warm_us": round(warm, 2), "cold_us": round(cold_us, 2), "desc": desc,
                              "cold_frac": round(b / (cold_us * 1e-6) / 8e12, 3), "warm_frac": round(b / (warm * 1e-6) / 8e12, 3)}
    ok = {k: v for k, v in out["paths"].items() if "cold_us" in v and k != "auto"}
    best = min(ok, key=lambda k: ok[k]["cold_us"])
    out["best_forced"] = best
    out["auto_over_best_cold"] = round(out["paths"]["auto"]["cold_us"] / ok[best]["cold_us"], 3)
    bestw = min(ok, key=lambda k: ok[k]["warm_us"])
    out["best_forced_warm"] = bestw
    out["auto_over_best_warm"] = round(out["paths"]["auto"]["warm_us"] / ok[bestw]["warm_us"], 3)
    return out


def table(points):
    names = [n for n, _ in PATHS]
    lines = ["| stand-in | band | far columns | " + " | ".join(f"{n} cold / warm us" for n in names) +
             " | auto = | best forced (cold) | auto / best cold | auto / best warm | auto cold frac | auto warm frac |",
             "|---|---|---|" + "---|" * (len(names) + 6)]
    for p in points:
        cells = []
        for n in names:
            v = p["paths"].get(n, {})
            cells.append(f"{v['cold_us']:.1f} / {v['warm_us']:.1f}" if "cold_us" in v else "n/a")
        a = p["paths"]["auto"]
        lines.append(f"| {p['kind']}{'' if p['kind'].startswith('rmat') else '-like'} | {p['band']:g} | {p['far']} | " + " | ".join(cells) +
                     f" | {a['desc']} | {p['best_forced']} | {p['auto_over_best_cold']:.3f} | {p['auto_over_best_warm']:.3f} | "
                     f"{a['cold_frac']:.3f} | {a['warm_frac']:.3f} |")
    return "\n".join(lines)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--scale", type=float, default=1.0)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--points", default="webbase,scircuit")
    ap.add_argument("--json", default=None)
    ap.add_argument("--md", default=None)
    args = ap.parse_args()
    dev = torch.device("cuda", 0)
    points = []
    for kind in args.points.split(","):
        for band, far in POINTS[kind]:
            p = run_point(kind, band, far, args.scale, dev, args.steps)
            points.append(p)
            print(json.dumps(p), flush=True)
    md = table(points)
    for kind in args.points.split(","):
        fr = [p["paths"]["auto"]["cold_frac"] for p in points if p["kind"] == kind]
        fw = [p["paths"]["auto"]["warm_frac"] for p in points if p["kind"] == kind]
        md += (f"\n\n**{kind}-like bracket (auto rules, over the points above)**: cold {min(fr):.3f} .. {max(fr):.3f} of the 8 TB/s roof, "
               f"warm {min(fw):.3f} .. {max(fw):.3f}.")
    bad = [p for p in points if p["auto_over_best_cold"] > 1.05]
    md += f"\n\nmis-picks of the auto rules (> 5 % over the best forced path, cold): {len(bad)} of {len(points)}" + \
          "".join(f"\n* {p['kind']} band {p['band']:g} {p['far']}: auto {p['paths']['auto']['cold_us']} us vs {p['best_forced']} "
                  f"{p['paths'][p['best_forced']]['cold_us']} us" for p in bad)
    print(md)
    if args.md:
        open(args.md, "w").write(md + "\n")
    if args.json:
        json.dump(points, open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Batch harness: run `bench.py --mtx` over Matrix Market files and collect one row per
matrix.  The reference's avx512 backend appends `file,GFlops` to results.csv (CSR5_avx512/main.cpp:105-110); this writes
the full JSON lines (`<out>.jsonl`) and a CSV with the roofline, conversion and CPU-baseline columns.

  python scripts/bench_batch.py matrices/*.mtx --out results --steps 200
  CSR5_MTX_DIR=/data/suitesparse python scripts/bench_batch.py --out results     # the files the BASELINE configs name

With no file arguments the SuiteSparse files of the BASELINE configs are taken from $CSR5_MTX_DIR (scircuit.mtx,
webbase-1M.mtx fp64; nd24k.mtx fp32) -- the same files bench.py substitutes for its synthetic stand-ins when present.
Small matrices: `roof_frac` / `us_per_spmv` are the COLD protocol's figures (bench.py roofline), `warm_frac` the
back-to-back one.
"""
import argparse
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COLUMNS = ["file", "m", "n", "nnz", "dtype", "sigma", "tiles", "csr_to_csr5_ms", "us_per_spmv", "gflops",
           "alg_GBps", "roof_frac", "warm_frac", "data", "cpu_gflops", "cpu_threads", "cpu_kind", "n_gpus", "ingest_parse_ms", "ingest_build_ms"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*", help=".mtx files or directories (default: the BASELINE files found in $CSR5_MTX_DIR)")
    ap.add_argument("--out", default="results", help="writes <out>.jsonl and <out>.csv (appends)")
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    files = []
    for f in args.files:
        files += sorted(glob.glob(os.path.join(f, "*.mtx"))) if os.path.isdir(f) else [f]
    dtype_of = {}
    if not files:
        sys.path.insert(0, ROOT)
        import bench
        for workload in ("scircuit", "webbase", "nd24k"):
            path = bench.real_file_for(workload)
            if path:
                files.append(path)
                dtype_of[path] = "f32" if workload == "nd24k" else "f64"  # BASELINE config 4 is the fp32 path
        if not files:
            raise SystemExit("bench_batch.py: no files given and none of scircuit.mtx / webbase-1M.mtx / nd24k.mtx in $CSR5_MTX_DIR")
    new_csv = not os.path.exists(args.out + ".csv")
    with open(args.out + ".jsonl", "a") as jl, open(args.out + ".csv", "a", newline="") as cf:
        w = csv.writer(cf)
        if new_csv:
            w.writerow(COLUMNS)
        for f in files:
            cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--mtx", f, "--steps", str(args.steps), "--warmup",
                   str(args.warmup), "--dtype", dtype_of.get(f, args.dtype), "--cpu-seconds", "2"]
            if args.no_cpu_baseline:
                cmd.append("--no-cpu-baseline")
            out = subprocess.run(cmd, capture_output=True, text=True, cwd=ROOT)
            line = next((l for l in out.stdout.splitlines() if l.startswith("{")), None)
            if out.returncode != 0 or line is None:
                print(f"{f}: failed ({out.returncode}) {out.stderr.strip().splitlines()[-1:]}", file=sys.stderr)
                continue
            d = json.loads(line)
            d["file"] = f
            jl.write(json.dumps(d) + "\n")
            c, r, b = d["config"], d["roofline"], d.get("cpu_baseline", {})
            ing = c.get("ingest_ms") or {}
            w.writerow([f, c["m_per_gpu"], c["n"], c["nnz_per_gpu"], d["dtype"], c["sigma"], c["tiles"], c["csr_to_csr5_ms"],
                        r["launch_us"], d["value"], r["achieved"], r["frac"], (r.get("warm") or {}).get("frac"), d.get("data"),
                        b.get("value"), b.get("cores"), b.get("kind"),
                        d["n_gpus"], ing.get("parse"), ing.get("coo_to_csr")])
            cf.flush()
            print(f"{os.path.basename(f):32s} nnz={c['nnz_per_gpu']:>10d} sigma={c['sigma']:2d} {r['launch_us']:9.2f} us "
                  f"{d['value']:8.1f} GFLOPS  {100 * r['frac']:5.1f} % of roof")


if __name__ == "__main__":
    main()

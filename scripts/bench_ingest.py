#!/usr/bin/env python3
"""Ingest benchmark: Matrix Market file -> CSR in HBM.

Writes a synthetic coordinate file (seeded), then times
  * cpu_baseline: the reference CLI's algorithm -- one fscanf per entry + serial counting scatter
    (oracle/mtx_oracle.c, main.cpp:181-275) on one host core, as the reference runs it;
  * ours: csr5hip_mtx_load = parallel mmap parse + H2D + device COO->CSR (stable radix sort + gather),
and checks that both CSRs are identical entry for entry.  Prints ONE JSON line.

usage: python scripts/bench_ingest.py [--entries 10000000] [--rows 1000000] [--symmetric] [--threads 0]
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def write_mtx(path, m, nz, symmetric, seed):
    rng = np.random.default_rng(seed)
    with open(path, "w") as f:
        f.write(f"%%MatrixMarket matrix coordinate real {'symmetric' if symmetric else 'general'}\n{m} {m} {nz}\n")
        step = 1 << 20
        for lo in range(0, nz, step):
            k = min(step, nz - lo)
            r = rng.integers(1, m + 1, k)
            c = rng.integers(1, m + 1, k)
            if symmetric:
                r, c = np.maximum(r, c), np.minimum(r, c)
            v = rng.standard_normal(k)
            np.savetxt(f, np.column_stack([r, c, v]), fmt="%d %d %.10g")
    return os.path.getsize(path)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--entries", type=int, default=10_000_000)
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--symmetric", action="store_true")
    ap.add_argument("--threads", type=int, default=0)
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--repeat", type=int, default=3)
    args = ap.parse_args()

    from benchmark_spmv_using_csr5_amd import _capi, ingest
    tmp = tempfile.mkdtemp(prefix="csr5_ingest_")
    path = os.path.join(tmp, "synthetic.mtx")
    t0 = time.time()
    size = write_mtx(path, args.rows, args.entries, args.symmetric, args.seed)
    gen_s = time.time() - t0

    best = None
    for _ in range(args.repeat):
        d = ingest.load_mtx(path, threads=args.threads)
        rec = dict(parse_ms=d.parse_ms, h2d_ms=d.h2d_ms, build_ms=d.build_ms, nnz=d.nnz)
        total = d.parse_ms + d.h2d_ms + d.build_ms
        if best is None or total < best[0]:
            best = (total, rec)
        host = d.to_host()
        d.release()
    total_ms, rec = best
    coo = ingest.read_mtx_coo(path, threads=args.threads)

    out = {
        "metric": "Matrix Market ingest (file -> CSR in HBM)", "unit": "M entries/s",
        "value": round(args.entries / total_ms * 1e-3, 2),
        "config": {"workload": f"synthetic real {'symmetric' if args.symmetric else 'general'} coordinate file",
                   "rows": args.rows, "entries": args.entries, "nnz": rec["nnz"], "file_MB": round(size / 1e6, 1),
                   "parser_threads": coo.threads, "fast_path": coo.fast_path},
        "phases_ms": {k: round(v, 3) for k, v in rec.items() if k.endswith("_ms")},
        "parse_GBps": round(size / rec["parse_ms"] * 1e-6, 3),
        "device_build": {"ms": round(rec["build_ms"], 3),
                         # COO in (16 B/entry) + CSR out (12 B/nnz) + row_ptr: the compulsory traffic
                         "algorithmic_GBps": round((16.0 * args.entries + 12.0 * rec["nnz"] + 4.0 * args.rows)
                                                   / rec["build_ms"] * 1e-6, 1)},
        "file_generation_s": round(gen_s, 1),
    }
    if not args.no_cpu_baseline:
        from oracle.csr5_oracle import Oracle, Reference
        if Reference.ingest_available():   # the reference CLI's own ingest (oracle/_ref, built from its sources)
            t0 = time.time()
            _, _, r_rp, r_col, r_val = Reference().ingest(path)
            cpu_s = time.time() - t0
            kind = "reference"
        else:                               # our C restatement of it
            t0 = time.time()
            seq = Oracle().mtx_read(path)
            cpu_s = time.time() - t0
            r_rp, r_col, r_val = seq.row_ptr, seq.col, seq.val
            kind = "port"
        same = (np.array_equal(r_rp, host.row_ptr) and np.array_equal(r_col, host.col)
                and np.array_equal(r_val.view(np.uint64), host.val.view(np.uint64)))
        out["cpu_baseline"] = {"value": round(args.entries / cpu_s * 1e-6, 3), "unit": "M entries/s", "cores": 1,
                               "kind": kind, "seconds": round(cpu_s, 2),
                               "sample": "same file, fscanf per entry + serial counting scatter (main.cpp:181-275)",
                               "csr_identical": bool(same)}
        out["speedup_vs_cpu"] = round(cpu_s * 1e3 / total_ms, 1)
    print(json.dumps(out))
    os.remove(path)
    os.rmdir(tmp)


if __name__ == "__main__":
    main()

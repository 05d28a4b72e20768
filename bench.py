#!/usr/bin/env python3
"""bench.py -- CSR5 SpMV throughput on N MI355X GPUs of one node (one process per GPU).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload scircuit|webbase|nd24k|rmat<S>]

A "step" is one SpMV pass (one ``spmv()`` call: y = A*x) over the rank's synthetic matrix shard, with every
input already resident in HBM.  At N = 1 the default workload is BASELINE.json configs[1] (SuiteSparse
scircuit, ~1 M nnz, fp64) as a seeded synthetic stand-in (no SuiteSparse files offline).  For N > 1 the
path shards by independent row blocks: every rank owns one row block of the same size
(weak scaling), x is replicated by ONE RCCL broadcast before the loop, and there is no per-step
collective.  value = 2 * (nnz over all ranks) * K / (max-over-ranks time of the K steps).

The JSON line also carries:
  roofline     -- algorithmic bytes per launch / HIP-event time per launch against the 8 TB/s HBM3E roof
  cpu_baseline -- the reference's own CSR5_avx2 (oracle/_ref, kind "reference") or our C port of it
                  (kind "port") timed on this node's host cores on the same matrix (rank 0, N = 1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured copy)


def profiled_traffic(key: str):
    """HBM-side bytes per launch of the dominant kernel from the committed rocprofv3 PMC passes
    (profiles/rNN_traffic.json, written by scripts/collect_profiles.py: (2*FETCH_SIZE + WRITE_SIZE) KiB,
    the factor 2 being the guide's gfx950 FETCH_SIZE correction).  Counters cannot be collected inside
    this process, so the value is looked up for exactly this workload/dtype/sigma/mode, else None."""
    import glob
    for path in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")), reverse=True):
        try:
            entry = json.load(open(path)).get(key)
        except Exception:
            entry = None
        if entry and entry.get("traffic_bytes_per_launch"):
            return int(entry["traffic_bytes_per_launch"]), os.path.basename(path)
    return None, None


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)   # reference NUM_RUN (CSR5_cuda/Makefile:5)
    ap.add_argument("--warmup", type=int, default=50)    # reference warm-up count (main.cu:85-89)
    ap.add_argument("--workload", default="scircuit")
    ap.add_argument("--mtx", default=None,
                    help="benchmark a Matrix Market file instead of a synthetic stand-in (single GPU): parsed and turned "
                         "into CSR by the native ingest, values replaced by rand()%%10 integers as the reference CLI does")
    ap.add_argument("--dtype", default=None, choices=[None, "f64", "f32"])
    ap.add_argument("--sigma", default="-1", help="-1 = auto rule (default), N = fixed, 'tuned' = measured autotune")
    ap.add_argument("--mode", default="fused", choices=["fused", "two-pass"])
    ap.add_argument("--launch", default="graph", choices=["graph", "eager"])
    ap.add_argument("--x-window", default="auto", choices=["auto", "off", "force"])
    ap.add_argument("--xcd-remap", type=int, default=1, choices=[0, 1])
    ap.add_argument("--lds-y", default="auto", choices=["auto", "off", "force"])
    ap.add_argument("--stream-nt", default="auto", choices=["auto", "off", "force"])
    ap.add_argument("--slabs", default="auto", help="column slabs: auto (default), 0 = off, 2..64 = that many")
    ap.add_argument("--slab-shift", type=int, default=None)
    ap.add_argument("--scaling", default="weak", choices=["weak", "strong"],
                    help="N > 1: weak = one fixed-size row block per GPU (default); strong = ONE global matrix "
                         "cut into nnz-balanced row blocks (BASELINE config: rmat24 over 8 GPUs)")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--scale", type=float, default=1.0, help="size factor of the synthetic stand-in (experiments)")
    ap.add_argument("--band", type=float, default=None,
                    help="scircuit / webbase: share of near-diagonal entries of the stand-in (defaults 0.5 / 0.3)")
    ap.add_argument("--spinup-seconds", type=float, default=0.0,
                    help="experiment knob: untimed replay of the same SpMV before the W warm-up steps (default off)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    return ap.parse_args()


def make_shard(workload: str, rank: int, world: int, seed: int, dtype, device, scale: float = 1.0,
               strong: bool = False, band=None):
    """Row block `rank` of a global matrix made of `world` equally sized row blocks.  Returns
    (CsrMatrix-like with device tensors or numpy arrays, global n)."""
    from benchmark_spmv_using_csr5_amd import matrices as M

    if workload.startswith("rmat"):
        scale = int(workload[4:] or 20)
        return (M.rmat_device(scale, 16, seed, rank, world, device, strong=strong),
                f"R-MAT scale {scale} EF16 (synthetic)")
    gen = {"scircuit": M.scircuit_like, "webbase": M.webbase_like, "nd24k": M.nd24k_like}[workload]
    kw = {} if scale == 1.0 else {"scale": scale}
    if workload in ("webbase", "scircuit") and band is not None:
        kw["band"] = band
    if workload == "scircuit" and os.environ.get("CSR5_BENCH_ROWCAP"):  # experiment knob, not a config
        kw["row_cap"] = int(os.environ["CSR5_BENCH_ROWCAP"])
    if strong and world > 1:  # one global matrix, nnz-balanced row blocks (sharding.py)
        from benchmark_spmv_using_csr5_amd import sharding as S
        full = gen(seed=seed, dtype=dtype, **kw)
        blk = S.extract_row_block(full.row_ptr, full.col, full.val, full.n,
                                  S.partition_rows_by_nnz(full.row_ptr, world), rank)
        return M.CsrMatrix(blk.m, blk.n, blk.row_ptr, blk.col, blk.val, full.name), full.name
    mat = gen(seed=seed + 101 * rank, dtype=dtype, **kw)
    if world > 1:  # spread the block's columns over the global column space of all blocks
        rng = np.random.default_rng(seed + 7 * rank)
        shift = rng.integers(0, world, size=mat.nnz, dtype=np.int64) * mat.n
        keep_local = rng.random(mat.nnz) < 0.5
        col = np.where(keep_local, mat.col.astype(np.int64) + rank * mat.n, mat.col.astype(np.int64) + shift)
        mat.col = col.astype(np.int32)
        mat.n = mat.n * world
    return mat, mat.name


def main():
    args = parse_args()
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a GPU"
    # CSR5_BENCH_SHARE_GPU=1 (test hook for 1-GPU boxes): every rank uses cuda:0 and the collectives go
    # through gloo, so the multi-rank control flow can be exercised without N GPUs.  Never set by default.
    share_gpu = os.environ.get("CSR5_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if share_gpu:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    from benchmark_spmv_using_csr5_amd import handle as H
    from benchmark_spmv_using_csr5_amd import matrices as M

    dtype_name = args.dtype or ("f32" if args.workload == "nd24k" else "f64")
    np_dtype = np.float64 if dtype_name == "f64" else np.float32
    t_dtype = torch.float64 if dtype_name == "f64" else torch.float32
    vsize = 8 if dtype_name == "f64" else 4

    ingest_ms = None
    if args.mtx:
        if world > 1:
            raise SystemExit("--mtx is a single-GPU option")
        from benchmark_spmv_using_csr5_amd import ingest
        loaded = ingest.load_mtx(args.mtx, dtype=np_dtype)
        ingest_ms = {"parse": round(loaded.parse_ms, 3), "h2d": round(loaded.h2d_ms, 3), "coo_to_csr": round(loaded.build_ms, 3)}
        mat = loaded.to_host(name=os.path.basename(args.mtx))
        loaded.release()
        label = f"{mat.name} (Matrix Market file, native ingest)"
    else:
        mat, label = make_shard(args.workload, rank, world, args.seed, np_dtype, dev, args.scale,
                                strong=args.scaling == "strong", band=args.band)
    if args.scale != 1.0:
        label += f" x{args.scale:g}"
    m, n, nnz = mat.m, mat.n, mat.nnz
    if isinstance(mat.row_ptr, np.ndarray):
        val, x_host = M.fill_values(nnz, n, np_dtype, seed=args.seed + 13, mode="int")
        rp = torch.from_numpy(mat.row_ptr).to(dev)
        ci = torch.from_numpy(mat.col).to(dev)
        va = torch.from_numpy(val).to(dev)
        xd = torch.from_numpy(x_host).to(dev)
    else:  # generated on the device
        rp, ci = mat.row_ptr, mat.col
        g = torch.Generator(device=dev).manual_seed(args.seed + 13 + rank)
        va = torch.randint(0, 10, (nnz,), generator=g, device=dev).to(t_dtype)
        xd = torch.randint(0, 10, (n,), generator=g, device=dev).to(t_dtype)
        val = x_host = None
    if world > 1:
        dist.broadcast(xd, src=0)  # the ONE collective: replicate x over xGMI (RCCL)
        x_host = xd.cpu().numpy()
    yd = torch.zeros(m, dtype=t_dtype, device=dev)

    A = H.anonymouslibHandle(m, n, dtype="float64" if dtype_name == "f64" else "float32")
    assert A.inputCSR(nnz, rp, ci, va) == 0
    assert A.setX(xd) == 0
    tuned = args.sigma == "tuned"
    assert A.setSigma(-1 if tuned else int(args.sigma)) == 0
    assert A.setSpmvMode(H.SPMV_FUSED if args.mode == "fused" else H.SPMV_TWO_PASS) == 0
    assert A.setXWindow({"off": 0, "auto": 1, "force": 2}[args.x_window]) == 0
    assert A.setOption(2, args.xcd_remap) == 0  # CSR5HIP_OPT_XCD_REMAP
    assert A.setLdsY({"off": 0, "auto": 1, "force": 2}[args.lds_y]) == 0
    assert A.setStreamNT({"off": 0, "auto": 1, "force": 2}[args.stream_nt]) == 0
    assert A.setColumnSlabs(1 if args.slabs == "auto" else int(args.slabs)) == 0
    if args.slab_shift is not None:
        assert A.setSlabShift(args.slab_shift) == 0
    A.warmup()
    torch.cuda.synchronize()
    if tuned:  # setup, outside every timed region (like asCSR5)
        err, _, _ = A.autotuneSigma(yd)
        assert err == 0, f"autotune -> {err}"
        assert A.asCSR() == 0
    t0 = time.perf_counter()
    err = A.asCSR5()
    torch.cuda.synchronize()
    convert_ms = (time.perf_counter() - t0) * 1e3
    assert err == 0, f"asCSR5 -> {err}"
    info = A.info()

    def run_steps(k: int):
        if k <= 0:
            return
        if args.launch == "graph":
            chunk = min(k, 500)
            for _ in range(k // chunk):
                assert A.spmv_repeat(1.0, yd, chunk) == 0
            if k % chunk:
                assert A.spmv_repeat(1.0, yd, k % chunk) == 0
        else:
            for _ in range(k):
                A.spmv(1.0, yd)

    # correctness run (kept for the cpu_baseline comparison), then W untimed warm-up steps
    assert A.spmv(1.0, yd) == 0
    torch.cuda.synchronize()
    y_first = yd.cpu().numpy() if rank == 0 else None
    if args.spinup_seconds > 0:
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < args.spinup_seconds:
            run_steps(500)
            torch.cuda.synchronize()
    run_steps(args.warmup)
    if args.launch == "graph":  # instantiate the graphs of the timed region outside it
        run_steps(args.steps)
    torch.cuda.synchronize()

    # ---- timed region: EXACTLY K steps, barrier + synchronize on both sides ----
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    A.timer_start()
    t0 = time.perf_counter()
    run_steps(args.steps)
    ev_ms = A.timer_stop()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    wall_s = time.perf_counter() - t0

    stats = torch.tensor([wall_s, ev_ms, float(nnz), float(m), float(n)], dtype=torch.float64, device=dev)
    if dist is not None:
        mx = stats.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        sm = stats.clone()
        dist.all_reduce(sm, op=dist.ReduceOp.SUM)
        wall_s, ev_ms = float(mx[0]), float(mx[1])
        total_nnz = float(sm[2])
    else:
        total_nnz = float(nnz)

    if rank == 0:
        ms_per_step = wall_s * 1e3 / args.steps
        gflops = 2.0 * total_nnz * args.steps / wall_s / 1e9
        b_alg = M.algorithmic_bytes(m, n, nnz, vsize)  # per launch, this rank's shard
        launch_ms = ev_ms / args.steps
        achieved = b_alg / (launch_ms * 1e-3) / 1e9
        traffic, traffic_src = (None, None)
        if world == 1:
            traffic, traffic_src = profiled_traffic(f"{label}|{dtype_name}|sigma={info.sigma}|{args.mode}")
        out = {
            "metric": f"{'fp64' if dtype_name == 'f64' else 'fp32'} SpMV GFLOPS",
            "value": round(gflops, 3),
            "unit": "GFLOPS",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 6),
            "higher_is_better": True,
            "scaling": args.scaling if world > 1 else "weak",
            "vs_baseline": None,
            "dtype": dtype_name,
            "data": "synthetic",
            "config": {
                "workload": f"{label}: CSR->CSR5 (omega=64, sigma={info.sigma}) + CSR5 SpMV, "
                            f"{'one row block per GPU, x replicated by one RCCL broadcast' if world > 1 else 'single GPU'}",
                "m_per_gpu": m, "n": n, "nnz_per_gpu": nnz, "sigma": info.sigma, "tiles": info.p,
                "spmv_mode": args.mode, "launch": args.launch,
                "lds_x_window": bool(info.x_window_active), "x_window_tiles": info.x_window_tiles,
                "x_window_cover_pct": info.x_window_cover_pct,
                "x_window_lines_per_gather": info.x_window_lines,
                "values": "rand()%10 integers (reference CLI data, exact in fp)",
                "clock_spinup_s": args.spinup_seconds,
                "ingest_ms": ingest_ms,
                "csr_to_csr5_ms": round(convert_ms, 3),
                "column_slabs": info.column_slabs, "slab_shift": info.slab_shift, "slab_segments": info.slab_segments,
                "slab_sigma": info.slab_sigma, "slab_build_ms": round(info.t_slab_ms, 3),
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 2),
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBS, 4),
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": "csr5::k_spmv",
                "algorithmic_bytes_per_launch": b_alg,
                # diagnostic: bytes the CSR5 kernel actually streams = B_alg with row_ptr replaced
                # by tile_ptr + tile_desc (x and y still counted once)
                "csr5_stream_bytes_per_launch": b_alg - 4 * (m + 1) + 4 * (info.p + 1)
                                                + 4 * info.p * 64 * info.num_packet,
                "launch_us": round(launch_ms * 1e3, 3),
            },
        }
        if world == 1 and not args.no_cpu_baseline and val is not None:
            out["cpu_baseline"] = cpu_baseline(mat, val, x_host, y_first, args.cpu_seconds)
        print(json.dumps(out), flush=True)

    A.destroy()
    A.close()
    if dist is not None:
        dist.destroy_process_group()


def cpu_baseline(mat, val, x, y_gpu, budget_s: float) -> dict:
    """CSR5 at omega=4 / sigma=16, fp64, OpenMP on this node's host cores, same matrix and vectors.
    Prefers the reference's own CSR5_avx2 build (oracle/_ref); falls back to our C port of it."""
    from oracle.csr5_oracle import Oracle, Reference

    m, n, nnz = mat.m, mat.n, mat.nnz
    val64 = val.astype(np.float64)
    x64 = x.astype(np.float64)
    if Reference.available():
        ref = Reference()
        # The reference runs with the ambient OpenMP thread count; on a 2-socket host the full count is far from
        # its best for a 1 M-nnz matrix, so give it the best of a few counts (short probe each), then time that one.
        full = ref.avx2_threads()
        tried = {}
        for t in sorted({full, max(full // 2, 1), max(full // 4, 1), max(full // 8, 1), 16, 8} - {0}):
            if t > full:
                continue
            ref.avx2_set_threads(t)
            _, ms_t, _ = ref.avx2_spmv(m, n, mat.row_ptr, mat.col, val64, x64, warm=5, runs=20)
            tried[t] = round(ms_t, 4)
        cores = min(tried, key=tried.get)
        ref.avx2_set_threads(cores)
        y, ms1, _ = ref.avx2_spmv(m, n, mat.row_ptr, mat.col, val64, x64, warm=2, runs=5)
        runs = int(max(20, min(2000, budget_s * 1e3 / max(ms1, 1e-3))))
        y, ms, conv_ms = ref.avx2_spmv(m, n, mat.row_ptr, mat.col, val64, x64, warm=50, runs=runs)
        kind = "reference"
    else:
        orc = Oracle()
        cores = orc.num_threads()
        fmt = orc.convert(4, 16, m, mat.row_ptr, mat.col, val64)
        t0 = time.perf_counter()
        y = orc.spmv(fmt, mat.row_ptr, x64)
        ms1 = (time.perf_counter() - t0) * 1e3
        runs = int(max(5, min(500, budget_s * 1e3 / max(ms1, 1e-3))))
        t0 = time.perf_counter()
        for _ in range(runs):
            y = orc.spmv(fmt, mat.row_ptr, x64)
        ms = (time.perf_counter() - t0) * 1e3 / runs
        conv_ms = None
        tried = {cores: round(ms, 4)}
        kind = "port"
    nonempty = np.diff(mat.row_ptr) > 0
    denom = np.maximum(np.abs(y[nonempty]), 1e-300)
    max_rel = float(np.max(np.abs(y_gpu[nonempty].astype(np.float64) - y[nonempty]) / denom)) if nonempty.any() else 0.0
    return {
        "value": round(2.0 * nnz / (ms * 1e-3) / 1e9, 3),
        "unit": "GFLOPS",
        "cores": cores,
        "kind": kind,
        "sample": f"same matrix ({nnz} nnz), CSR5_avx2 omega=4 sigma=16 fp64 OpenMP, 50 warm-up + {runs} timed SpMV",
        # 20-run probes per thread count (ms per SpMV).  They can be several times faster than the long timed run:
        # the box's container throttles sustained multi-thread CPU use, short bursts escape it.
        "threads_tried_ms": tried,
        "burst_gflops": round(2.0 * nnz / (min(tried.values()) * 1e-3) / 1e9, 3),
        "ms_per_spmv": round(ms, 5),
        "csr_to_csr5_ms": None if conv_ms is None else round(conv_ms, 3),
        "max_rel_err_gpu_vs_cpu": max_rel,
    }


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Copy the rocprofv3 summaries produced by scripts/gpu_profiles.sh from gpurun_out/ (scratch) into
profiles/ (tracked) and derive the per-launch HBM traffic that bench.py reports as roofline.traffic.

traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes per launch: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950
FETCH_SIZE counts 128-B fabric requests as 64 B (MI355X_MICROARCH.md, HBM section), hence the factor 2.
Calibration in our own access pattern (4-/8-byte-per-lane coalesced wave loads): the nd24k-like fp32 matrix
is almost pure streaming and 2*FETCH_SIZE comes out at 1.01-1.03 x its algorithmic bytes.
usage: python scripts/collect_profiles.py r01
"""
import collections, csv, glob, json, os, shutil, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
rnd = sys.argv[1] if len(sys.argv) > 1 else "r01"
src = os.path.join(ROOT, "gpurun_out")
dst = os.path.join(ROOT, "profiles")
os.makedirs(dst, exist_ok=True)
traffic = {}
rows = []
ingest = []
for d in sorted(glob.glob(os.path.join(src, "profiles_*"))):
    tag = os.path.basename(d)[len("profiles_"):]
    stats = os.path.join(d, "trace", "t_kernel_stats.csv")
    if not os.path.exists(stats):
        continue
    shutil.copy(stats, os.path.join(dst, f"{rnd}_{tag}_kernel_stats.csv"))
    line = json.loads(open(os.path.join(d, "bench_line.json")).read())
    if tag.startswith("ingest_"):
        top = sorted(csv.DictReader(open(stats)), key=lambda r: -float(r["TotalDurationNs"]))[:6]
        ingest.append((tag, line, top))
        continue
    pmc = collections.defaultdict(list)
    for f in glob.glob(os.path.join(d, "pmc_*", "p_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "k_spmv" in r["Kernel_Name"]:
                pmc[r["Counter_Name"]].append(float(r["Counter_Value"]))
    avg = {k: sum(v) / len(v) for k, v in pmc.items()}
    spmv = [r for r in csv.DictReader(open(stats)) if "k_spmv" in r["Name"]]
    cal = [r for r in csv.DictReader(open(stats)) if "k_calibrate" in r["Name"]]
    t = {"fetch_size_kib": avg.get("FETCH_SIZE"), "write_size_kib": avg.get("WRITE_SIZE"),
         "tcc_hit": avg.get("TCC_HIT_sum"), "tcc_miss": avg.get("TCC_MISS_sum")}
    if t["fetch_size_kib"] is not None and t["write_size_kib"] is not None:
        t["traffic_bytes_per_launch"] = int((2 * t["fetch_size_kib"] + t["write_size_kib"]) * 1024)
    key = f"{line['config']['workload'].split(':')[0]}|{line['dtype']}|sigma={line['config']['sigma']}|{line['config']['spmv_mode']}"
    t.update(key=key, k_spmv_avg_ns=float(spmv[0]["AverageNs"]) if spmv else None,
             k_spmv_min_ns=float(spmv[0]["MinNs"]) if spmv else None, k_spmv_calls=int(spmv[0]["Calls"]) if spmv else 0,
             k_calibrate_avg_ns=float(cal[0]["AverageNs"]) if cal else None,
             bench_under_rocprof=line)
    traffic[key] = t
    b = line["roofline"]["algorithmic_bytes_per_launch"]
    rows.append((tag, line["config"]["sigma"], line["config"]["spmv_mode"], t["k_spmv_avg_ns"], t["k_spmv_min_ns"],
                 line["roofline"]["launch_us"], b, t.get("traffic_bytes_per_launch"),
                 None if not t["tcc_hit"] else t["tcc_hit"] / (t["tcc_hit"] + t["tcc_miss"])))
json.dump(traffic, open(os.path.join(dst, f"{rnd}_traffic.json"), "w"), indent=1)
with open(os.path.join(dst, f"{rnd}_summary.md"), "w") as f:
    f.write(f"# {rnd}: rocprofv3 summaries (MI355X, `scripts/gpu_profiles.sh`)\n\n"
            "`*_kernel_stats.csv` = `rocprofv3 --kernel-trace --stats` of `python bench.py --no-cpu-baseline <args>`;\n"
            "traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB from separate `--pmc` passes (see scripts/collect_profiles.py).\n\n"
            "| run | sigma | mode | k_spmv avg us (rocprof) | k_spmv min us | HIP-event us/launch (bench.py, same run) | B_alg MB | HBM-side traffic MB | L2 hit |\n|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        f.write("| %s | %d | %s | %.2f | %.2f | %.2f | %.1f | %s | %s |\n" % (
            r[0], r[1], r[2], r[3] / 1e3, r[4] / 1e3, r[5], r[6] / 1e6,
            "n/a" if r[7] is None else "%.1f" % (r[7] / 1e6), "n/a" if r[8] is None else "%.0f %%" % (100 * r[8])))
    if ingest:
        f.write("\n## Matrix Market ingest (`scripts/bench_ingest.py` under `rocprofv3 --kernel-trace --stats`)\n\n"
                "| run | entries | nnz | parse ms | H2D ms | device COO->CSR ms | M entries/s | reference algorithm on 1 host core, s |\n|---|---|---|---|---|---|---|---|\n")
        for tag, line, top in ingest:
            ph, cfg = line["phases_ms"], line["config"]
            f.write("| %s | %d | %d | %.1f | %.1f | %.2f | %.0f | %s |\n" % (
                tag, cfg["entries"], cfg["nnz"], ph["parse_ms"], ph["h2d_ms"], ph["build_ms"], line["value"],
                line.get("cpu_baseline", {}).get("seconds", "n/a")))
        for tag, line, top in ingest:
            f.write(f"\nLongest device kernels of `{tag}` (all calls of the run, ns):\n\n| kernel | calls | total | average |\n|---|---|---|---|\n")
            for r in top:
                f.write("| `%s` | %s | %s | %.0f |\n" % (r["Name"][:90], r["Calls"], r["TotalDurationNs"], float(r["AverageNs"])))
print(open(os.path.join(dst, f"{rnd}_summary.md")).read())

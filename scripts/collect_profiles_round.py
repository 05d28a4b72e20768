#!/usr/bin/env python3
"""gpurun_out/profiles_<round>/ (scratch, written by scripts/gpu_profiles_round.sh on the GPU box) -> profiles/<round>_* (tracked):
kernel-stats csv per workload, <round>_traffic.json (what bench.py looks roofline.traffic up in; every entry carries the hash
of the kernel sources it was measured on) and <round>_summary.md.  usage: collect_profiles_round.py [r03]"""
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (kernel_source_hash)

ROUND = sys.argv[1] if len(sys.argv) > 1 else "r03"
src = os.path.join(ROOT, "gpurun_out", f"profiles_{ROUND}")
dst = os.path.join(ROOT, "profiles")
traffic, rows = {}, []
for f in sorted(glob.glob(os.path.join(src, "*_pmc.json"))):
    tag = os.path.basename(f)[:-len("_pmc.json")]
    pmc = json.load(open(f))
    line = json.loads(open(os.path.join(src, f"{tag}_bench_line.json")).read())
    shutil.copy(os.path.join(src, f"{tag}_kernel_stats.csv"), os.path.join(dst, f"{ROUND}_{tag}_kernel_stats.csv"))
    cfg, roof = line["config"], line["roofline"]
    label = cfg["workload"].split(":")[0]
    key = (f"{label}|{line['dtype']}|sigma={cfg['sigma']}|{cfg['spmv_mode']}|slabs={cfg['column_slabs']}/{cfg['slab_shift']}"
           f"/hot={int(cfg['slab_hot_table'])}")
    t = pmc.get("traffic_bytes_per_step")
    step_ns = sum(k["avg_ns"] * k["calls"] for k in pmc.get("kernel_stats", []))
    main = max(pmc.get("kernel_stats", [{"calls": 0, "avg_ns": 0, "name": ""}]), key=lambda k: k["avg_ns"] * k["calls"])
    steps = main["calls"]
    entry = {"key": key, "kernel_source_hash": bench.kernel_source_hash(), "traffic_bytes_per_launch": t,
             "fetch_size_kib_per_step": pmc.get("FETCH_SIZE_per_step"), "write_size_kib_per_step": pmc.get("WRITE_SIZE_per_step"),
             "tcc_hit_per_step": pmc.get("TCC_HIT_sum_per_step"), "tcc_miss_per_step": pmc.get("TCC_MISS_sum_per_step"),
             "kernel_stats": pmc.get("kernel_stats"), "rocprof_step_us": step_ns / steps / 1e3 if steps else None,
             "bench_under_rocprof": line}
    traffic[key] = entry
    rows.append((tag, cfg, roof, entry, main))
json.dump(traffic, open(os.path.join(dst, f"{ROUND}_traffic.json"), "w"), indent=1)
with open(os.path.join(dst, f"{ROUND}_summary.md"), "w") as f:
    f.write(f"# {ROUND}: rocprofv3 summaries (MI355X, `scripts/gpu_profiles_round.sh`)\n\n"
            f"`{ROUND}_<run>_kernel_stats.csv` = `rocprofv3 --kernel-trace --stats` of `python bench.py --no-cpu-baseline --no-sub-configs <args>`.\n"
            "Step time (rocprof) = sum over the step's kernels (tile kernel [+ tail / range-finish launch] [+ slab combine]) of average duration.\n"
            "traffic = (2*FETCH_SIZE + WRITE_SIZE) KiB per step from separate `--pmc` passes (counters only).\n"
            "Small workloads run with `--no-cold` here (one protocol per trace: cache-warm back-to-back steps); their HIP-event time under the\n"
            "profiler carries its per-dispatch overhead (6-40 us kernels: +2 ... +14 us), the kernel's own average duration does not.\n\n"
            "| run | sigma | slabs / hot | dominant kernel | its avg us | step us (rocprof, all kernels) | HIP-event us/step (bench, same run) | B_alg MB | traffic MB | traffic / B_alg | frac of 8 TB/s (step, rocprof) | L2 hit |\n"
            "|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for tag, cfg, roof, e, main in rows:
        b = roof["algorithmic_bytes_per_launch"]
        t = e["traffic_bytes_per_launch"]
        hit = e["tcc_hit_per_step"]
        miss = e["tcc_miss_per_step"]
        step = e["rocprof_step_us"]
        f.write("| %s | %d | %d / %d | `%s` | %.2f | %.2f | %.2f | %.1f | %s | %s | %.3f | %s |\n" % (
            tag, cfg["sigma"], cfg["column_slabs"], int(cfg["slab_hot_table"]), main["name"].replace("void ", "")[:48],
            main["avg_ns"] / 1e3, step, roof["launch_us"], b / 1e6, "n/a" if t is None else "%.1f" % (t / 1e6),
            "n/a" if t is None else "%.2f" % (t / b), b / (step * 1e-6) / 8e12 if step else 0,
            "n/a" if not hit else "%.0f %%" % (100 * hit / (hit + miss))))
tail = os.path.join(dst, f"{ROUND}_summary_tail.md")  # hand-written deltas of the round, kept across re-collections
if os.path.isfile(tail):
    with open(os.path.join(dst, f"{ROUND}_summary.md"), "a") as f:
        f.write(open(tail).read())
print(open(os.path.join(dst, f"{ROUND}_summary.md")).read())

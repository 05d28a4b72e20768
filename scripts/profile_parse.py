#!/usr/bin/env python3
"""Runs on the GPU box (scripts/gpu_profiles.sh): condenses one workload's rocprofv3 output into <tag>_pmc.json.

Per SpMV step = every dispatch of the step's kernels (tile kernel, its tail-only launch, slab combine, calibrate).
traffic = (2 * FETCH_SIZE + WRITE_SIZE) KiB summed over those dispatches / number of steps: FETCH_SIZE / WRITE_SIZE are
in KiB and on gfx950 FETCH_SIZE counts a 128-byte fabric request as 64 bytes (MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import glob
import json
import os
import sys

tag, d, out = sys.argv[1], sys.argv[2], sys.argv[3]
STEP = ("k_spmv", "k_slab_combine", "k_calibrate", "k_range_finish", "k_x_permute")
MAIN = ("k_spmv_hot", "k_spmv_range", "k_spmv<")


def is_step(name):
    return any(k in name for k in STEP)


res = {"tag": tag}
sums = collections.defaultdict(float)
per_kernel = collections.defaultdict(lambda: collections.defaultdict(float))
steps = 0
for f in glob.glob(os.path.join(d, "pmc_*", "**", "*counter_collection.csv"), recursive=True):
    rows = list(csv.DictReader(open(f)))
    names = collections.Counter()
    for r in rows:
        k = r.get("Kernel_Name", "")
        if not is_step(k):
            continue
        sums[r["Counter_Name"]] += float(r["Counter_Value"])
        per_kernel[k.split("(")[0][:70]][r["Counter_Name"]] += float(r["Counter_Value"])
        names[(k.split("(")[0], r["Counter_Name"])] += 1
    # steps of this pass = dispatches of the most frequent main kernel (one per step)
    mains = [v for (k, c), v in names.items() if any(m in k for m in MAIN)]
    n = max(mains) if mains else 0
    for (k, c), v in names.items():
        res.setdefault("dispatches", {})[c] = max(res.get("dispatches", {}).get(c, 0), n)
for c, total in sums.items():
    n = res["dispatches"].get(c, 0)
    res[c + "_per_step"] = total / n if n else None
if res.get("FETCH_SIZE_per_step") is not None and res.get("WRITE_SIZE_per_step") is not None:
    res["traffic_bytes_per_step"] = int((2 * res["FETCH_SIZE_per_step"] + res["WRITE_SIZE_per_step"]) * 1024)
res["per_kernel_totals"] = {k: dict(v) for k, v in per_kernel.items()}
stats = glob.glob(os.path.join(out, f"{tag}_kernel_stats.csv"))
if stats:
    ks = [r for r in csv.DictReader(open(stats[0])) if is_step(r["Name"])]
    res["kernel_stats"] = [{"name": r["Name"].split("(")[0][:80], "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"]),
                            "min_ns": float(r["MinNs"]), "max_ns": float(r["MaxNs"])} for r in ks]
json.dump(res, open(os.path.join(out, f"{tag}_pmc.json"), "w"), indent=1)
print(tag, {k: v for k, v in res.items() if k.endswith("per_step")})

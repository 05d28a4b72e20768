#!/usr/bin/env python3
"""Gaps between consecutive kernels of a rocprofv3 --kernel-trace csv: average (end -> next start) per kernel pair.

    python scripts/experiments/kernel_gaps.py <..._kernel_trace.csv>
"""
import collections
import csv
import sys

rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
short = lambda n: n.split("(")[0].replace("void csr5::", "").split("<")[0]
gaps, durs = collections.defaultdict(list), collections.defaultdict(list)
for a, b in zip(rows, rows[1:]):
    gaps[(short(a["Kernel_Name"]), short(b["Kernel_Name"]))].append(int(b["Start_Timestamp"]) - int(a["End_Timestamp"]))
for r in rows:
    durs[short(r["Kernel_Name"])].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
for k, v in sorted(durs.items(), key=lambda kv: -sum(kv[1])):
    if len(v) >= 10:
        v = sorted(v)
        print(f"kernel {k:28s} n={len(v):5d} median {v[len(v)//2]/1e3:9.2f} us")
for k, v in sorted(gaps.items(), key=lambda kv: -len(kv[1])):
    if len(v) >= 10:
        v = sorted(v)
        print(f"gap {k[0]:24s} -> {k[1]:24s} n={len(v):5d} median {v[len(v)//2]/1e3:8.2f} us")

#!/usr/bin/env python3
"""Timeline of the LAST CSR->CSR5 conversion in a rocprofv3 --kernel-trace csv: start offset, duration, gap, kernel.
usage: conv_trace.py kernel_trace.csv [kernels_to_show]"""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
show = int(sys.argv[2]) if len(sys.argv) > 2 else 14
# the parent's conversion starts with the arena memset before the second-to-last k_row_scan when a slab child exists
scans = [i for i, r in enumerate(rows) if "k_row_scan" in r["Kernel_Name"]]
idx = scans[-1] if show <= 14 or len(scans) < 2 else scans[-2]
t0 = int(rows[idx - 1]["Start_Timestamp"])
prev_end = None
for r in rows[idx - 1: idx - 1 + show]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%9.2f  dur %8.2f  gap %7.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:80]))
    prev_end = e

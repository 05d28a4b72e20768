#!/usr/bin/env python3
"""Timeline of the LAST CSR->CSR5 conversion in a rocprofv3 --kernel-trace csv: start offset, duration, kernel."""
import csv
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
idx = [i for i, r in enumerate(rows) if "k_row_scan" in r["Kernel_Name"]][-1]
t0 = int(rows[idx - 1]["Start_Timestamp"])
prev_end = None
for r in rows[idx - 2: idx + 12]:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    gap = (s - prev_end) / 1e3 if prev_end else 0.0
    print("%8.2f  dur %7.2f  gap %6.2f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, gap, r["Kernel_Name"][:70]))
    prev_end = e

#!/usr/bin/env python3
"""Compact one-line view of a bench.py JSON line read from stdin (experiment scripts)."""
import json, sys
for line in sys.stdin:
    line = line.strip()
    if not line.startswith("{"):
        continue
    d = json.loads(line)
    c, r = d["config"], d["roofline"]
    print(f"{c['workload'][:26]:26s} {d['dtype']} sigma={c['sigma']:2d} {c['spmv_mode']:8s} xwin={int(c.get('lds_x_window', 0))}"
          f" cover={c.get('x_window_cover_pct', 0):3d}% tiles={c['tiles']:6d}  {d['value']:9.1f} GFLOPS  {r['launch_us']:9.3f} us"
          f"  {r['achieved']:7.1f} GB/s  frac={r['frac']:.3f}  slabs={c.get('column_slabs', 0)}/{c.get('slab_shift', 0)} hot={int(c.get('slab_hot_table', 0))}/{c.get('slab_hot_cover_pct', 0)}%"
          f" seg={c.get('slab_segments', 0)} conv={c.get('csr_to_csr5_ms', 0)}ms slabbuild={c.get('slab_build_ms', 0)}ms")

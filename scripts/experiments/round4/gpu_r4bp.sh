#!/bin/bash
# round 4, call bp: TIMING ONLY -- k_spmv's column / value streams fetched with 16-byte loads (elements in the wrong lanes: wrong results);
# is the one-tile kernel of the small configs sensitive to the number of its load instructions?
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; w = r.get('warm', {}); c = d['config']
        print('%-22s sigma %2d cold %8.2f us frac %.3f | warm %8.2f us frac %.3f' % (c['workload'][:22], c['sigma'], r['launch_us'], r['frac'], w.get('launch_us', 0), w.get('frac', 0)))
"; }
for w in nd24k "nd24k --dtype f64"; do for v in base abl_widespmv base abl_widespmv; do echo -n "$v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so python bench.py --no-cpu-baseline --no-sub-configs --workload $w 2>&1 | tail -1 | line; done; done
for v in base abl_widespmv; do echo -n "$v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so python bench.py --no-cpu-baseline --no-sub-configs --workload webbase --sigma 8 2>&1 | tail -1 | line; done
for v in base abl_widespmv; do echo -n "$v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so python bench.py --no-cpu-baseline --no-sub-configs --workload scircuit --sigma 8 2>&1 | tail -1 | line; done

#!/bin/bash
# round 4, call br: the x-window of the one-tile kernel staged with 16-byte loads / LDS stores (4 + 4 instead of 16 + 16 per tile): parity, pairs
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -3
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; w = r.get('warm', {}); c = d['config']
        print('%-22s %s xwin %d cover %3d%% cold %8.2f us frac %.3f | warm %8.2f us frac %.3f' % (c['workload'][:22], d['dtype'], int(c.get('lds_x_window', 0)), c.get('x_window_cover_pct', 0), r['launch_us'], r['frac'], w.get('launch_us', 0), w.get('frac', 0)))
"; }
for w in nd24k "nd24k --dtype f64"; do for v in base xwide base xwide base xwide; do echo -n "$v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so python bench.py --no-cpu-baseline --no-sub-configs --workload $w 2>&1 | tail -1 | line; done; done

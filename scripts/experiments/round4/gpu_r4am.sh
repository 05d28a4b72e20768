#!/bin/bash
# round 4, call am: k_spmv touches the streams of the tile one resident set ahead (CSR5_PREFETCH = percent of a resident set);
# cold and warm step of the three small configs, each value of the knob twice
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; w = r.get('warm', {})
        print('%-28s cold %8.2f us frac %.3f | warm %8.2f us frac %.3f' % (d['config']['workload'][:28], r['launch_us'], r['frac'], w.get('launch_us', 0), w.get('frac', 0)))
"; }
for w in nd24k webbase scircuit; do
  for p in 0 50 100 200 0 100; do
    echo -n "prefetch $p: "; CSR5_PREFETCH=$p timeout 300 python bench.py --no-cpu-baseline --no-sub-configs --workload $w 2>&1 | tail -1 | line
  done
done

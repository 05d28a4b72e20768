#!/bin/bash
# round 4, call bf: the small power-law configs with a forced hot table on the round-4 range kernel (round 3 measured them on its kernel)
line() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); r = d['roofline']; w = r.get('warm', {}); c = d['config']
        print('%-22s slabs %2s hot %d/%2d%% cold %8.2f us frac %.3f | warm %8.2f us frac %.3f' % (c['workload'][:22], c.get('column_slabs'), int(c.get('slab_hot_table', 0)), c.get('slab_hot_cover_pct', 0), r['launch_us'], r['frac'], w.get('launch_us', 0), w.get('frac', 0)))
"; }
for w in webbase scircuit; do
  python bench.py --no-cpu-baseline --no-sub-configs --workload $w 2>&1 | tail -1 | line
  for s in 8 16; do
    python bench.py --no-cpu-baseline --no-sub-configs --workload $w --slabs $s --slab-hot force 2>&1 | tail -1 | line
  done
done

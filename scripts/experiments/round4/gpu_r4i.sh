#!/bin/bash
# round 4, call i: tiles per wavefront of the x-window kernel (nd24k-like fp32 and fp64), warm and cold
cold() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('   cold %.2f us frac %.3f | warm %.2f us frac %.3f | sigma %d xwin %s tpw %s' % (r['launch_us'], r['frac'], r['warm']['launch_us'], r['warm']['frac'], d['config']['sigma'], d['config']['lds_x_window'], d['config'].get('tiles_per_wave')))"; }
for rep in 1 2; do
for t in 1 2 3 4; do echo "== nd24k f32 tpw $t"; cold --workload nd24k --tiles-per-wave $t; done
done
for t in 1 2 4; do echo "== nd24k f64 tpw $t"; cold --workload nd24k --dtype f64 --tiles-per-wave $t; done
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -k "window or xwin or nd24k or zoo" 2>&1 | tail -3

#!/bin/bash
# round 4, call bn: lane-major 16-byte pieces of the hot child's values, more same-call pairs on R-MAT 24 (headline and narrowed)
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(d['value'], r['launch_us'], r['frac'], 'narrowed', (r.get('narrowed_values') or {}).get('launch_us'))"; }
for v in base pieces base pieces base pieces base pieces base pieces; do echo -n "rmat24 $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload rmat24; done

#!/bin/bash
# round 4, call bo: the fp32 copy of the values (CSR5HIP_OPT_NARROW_VALUES) in lane-major pieces of four floats: parity, then same-call pairs
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -k "narrow or hot" 2>&1 | grep -E "passed|failed|rror|assert" | tail -3
CSR5_FUZZ_SEED=818 CSR5_FUZZ_CASES=1000 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(d['value'], r['launch_us'], r['frac'], 'narrowed', (r.get('narrowed_values') or {}).get('launch_us'))"; }
for v in base pieces base pieces base pieces base pieces; do echo -n "rmat24 $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload rmat24; done
for v in base pieces base pieces; do echo -n "rmat22 $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload rmat22; done

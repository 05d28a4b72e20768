#!/bin/bash
# round 4, call bm: the hot child's values in lane-major 16-byte pieces (half the value load instructions): parity, then same-call pairs
timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | grep "smoke" | tail -3
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
CSR5_FUZZ_SEED=717 CSR5_FUZZ_CASES=1500 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(d['value'], r['launch_us'], r['frac'], 'narrowed', (r.get('narrowed_values') or {}).get('launch_us'), 'conv', d['config'].get('csr_to_csr5_ms'))"; }
for w in rmat24 rmat22; do for v in base pieces base pieces base pieces; do echo -n "$w $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload $w; done; done

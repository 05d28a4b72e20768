#!/bin/bash
# round 4, call ay: CSR5HIP_OPT_NARROW_VALUES (fp32-exact fp64 values streamed as fp32): parity and the side figure of the bench
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -5
for w in rmat24 rmat22; do python bench.py --no-cpu-baseline --no-sub-configs --workload $w 2>&1 | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.readline()); r = d['roofline']
print(d['config']['workload'][:20], d['value'], r['launch_us'], r['frac'], json.dumps(r.get('narrowed_values'))[:400])"; done

#!/bin/bash
# same-call A/B of library builds on the headline workload: step time of `bench.py` (live x) per library; usage: step_ab.sh lib1 lib2 ...
# ("product" = the in-tree build)
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = product ]; then unset CSR5HIP_LIB; else export CSR5HIP_LIB=scripts/probes/libcsr5hip_$lib.so; fi
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-sub-configs --no-side-figures ${BENCH_ARGS:-} 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('$lib', 'us/step', round(d['event_ms_per_step']*1e3,1), 'frac', d['roofline']['frac'])"
done
done

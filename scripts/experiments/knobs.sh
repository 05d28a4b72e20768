#!/bin/bash
# re-check the auto decisions (LDS y segments, LDS x-window, XCD remap) on the final kernels, 3 processes per point
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py | awk '{print $(NF-4)}'; }
for w in "" "--workload webbase --steps 300" "--workload nd24k --steps 200" "--workload rmat20 --steps 100 --warmup 5"; do
  for k in "" "--lds-y off" "--lds-y force" "--x-window force" "--x-window off" "--xcd-remap 0"; do
    echo -n "[$w] [$k]: "; for i in 1 2 3; do one $w $k; done | tr '\n' ' '; echo
  done
done

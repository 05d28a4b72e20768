#!/bin/bash
# A/B of library builds INSIDE one gpurun call (boxes differ by +-3 %: never compare across calls).
# usage: ab_libs.sh "<workloads>" <libA.so> <libB.so> ...   (paths relative to the repo root)
wl=$1; shift
for rep in 1 2; do
  for lib in "$@"; do
    for w in $wl; do
      CSR5HIP_LIB=$PWD/$lib timeout 300 python bench.py --workload $w --no-cpu-baseline --no-sub-configs 2>/dev/null | python scripts/benchline.py | awk -v l=$lib '{print l, $0}' | cut -c1-190
    done
  done
done

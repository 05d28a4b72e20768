#!/bin/bash
# round 4, call j: wavefronts per CU of the persistent kernel vs table size (16 waves / 12 288 slots, 12 / 14 336, 8 / 16 384)
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w 16 waves"; one --workload $w
    echo "== $w 12 waves"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w12.so one --workload $w
    echo "== $w 8 waves"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w8.so one --workload $w
  done
done

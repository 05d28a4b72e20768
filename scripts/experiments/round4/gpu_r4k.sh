#!/bin/bash
# round 4, call k: fewer wavefronts per CU, larger table (8 waves / 16 384 slots, 6 / 17 408, 4 / 18 432)
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w 8 waves"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w8.so one --workload $w
    echo "== $w 6 waves"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w6.so one --workload $w
    echo "== $w 4 waves"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w4.so one --workload $w
  done
done

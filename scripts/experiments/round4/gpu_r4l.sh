#!/bin/bash
# round 4, call l: prefetch depth 3 (streams two tiles ahead) at 8 and 16 wavefronts per CU
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w 8 waves depth 2"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w8.so one --workload $w
    echo "== $w 8 waves depth 3"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w8d3.so one --workload $w
    echo "== $w 16 waves depth 3"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w16d3.so one --workload $w
  done
done

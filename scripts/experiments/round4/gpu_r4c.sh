#!/bin/bash
# round 4, call c: same-call A/B of issue-order / scalar tile_ptr variants of k_spmv_range, and child sigma 4
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w base"; one --workload $w
    echo "== $w tpscalar"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_tpscalar.so one --workload $w
    echo "== $w streams-first"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_sfirst.so one --workload $w
  done
done
echo "== rmat24 sigma 4"; one --workload rmat24 --sigma 4
echo "== rmat22 sigma 4"; one --workload rmat22 --sigma 4

#!/bin/bash
# round 4, call ac: child sigma 8 / 12 / 16 at 8 wavefronts per CU (a larger tile amortises the per-tile work that two wavefronts
# per SIMD no longer hide; its y region takes LDS from the table: 16 384 / 14 336 / 12 288 slots), and sigma 16 at 6 wavefronts
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w sigma 8"; one --workload $w
    echo "== $w sigma 12"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_s12.so one --workload $w
    echo "== $w sigma 16"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_s16.so one --workload $w
    echo "== $w sigma 16, 6 waves"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_s16w6.so one --workload $w
  done
done

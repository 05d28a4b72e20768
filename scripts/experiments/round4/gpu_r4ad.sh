#!/bin/bash
# round 4, call ad: fp64 child sigma 16 / 12 / 8 with the SAME 16 384-slot table (4-KB y region, a tile of very short rows walks its
# flags in windows), 8 wavefronts per CU; parity first
timeout 1200 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w sigma 8"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_s8.so one --workload $w
    echo "== $w sigma 12"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_s12w.so one --workload $w
    echo "== $w sigma 16"; one --workload $w
  done
done

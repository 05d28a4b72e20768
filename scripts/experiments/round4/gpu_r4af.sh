#!/bin/bash
# round 4, call af: 16 (12) wavefronts per CU taking turns on 8 (6) y-compaction regions (LDS lock): occupancy of the round-3
# geometry with the table of the round-4 one
CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w16sh2.so timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w16sh2.so CSR5_FUZZ_SEED=77 CSR5_FUZZ_CASES=1500 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w 8 waves (product)"; one --workload $w
    echo "== $w 16 waves on 8 regions"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w16sh2.so one --workload $w
    echo "== $w 12 waves on 6 regions"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_w12sh2.so one --workload $w
  done
done

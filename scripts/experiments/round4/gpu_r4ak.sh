#!/bin/bash
# round 4, call ak: the child's bit flags ride in bit 22 of the column codes, y_offset is recomputed in the kernel, the descriptor
# array is not read (-256 B of 6 144 B per tile): parity, fuzz, A/B vs the previous library
timeout 1200 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py tests/test_gpu_multi.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
CSR5_FUZZ_SEED=611 CSR5_FUZZ_CASES=3000 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
CSR5_FUZZ_SCALE=30 CSR5_FUZZ_SEED=612 CSR5_FUZZ_CASES=600 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2 3; do
  for w in rmat24 rmat22; do
    echo "== $w before"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_desc.so one --workload $w
    echo "== $w after"; one --workload $w
  done
done

#!/bin/bash
# round 4, call al: whole gpu suite after dropping the hot child's descriptor array; conversion time; bench
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
CSR5_FUZZ_SEED=613 CSR5_FUZZ_CASES=3000 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230; }
for w in rmat24 rmat22; do echo "== $w before"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_desc.so one --workload $w; echo "== $w after"; one --workload $w; done
timeout 600 python scripts/experiments/scale_check.py --scale 25 2>&1 | tail -1

#!/bin/bash
# round 4, call v: combine with packed row bytes in its first round (A/B vs the previous library), parity
timeout 1200 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22 webbase; do
    echo "== $w before"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_cmb0.so one --workload $w
    echo "== $w after"; one --workload $w
  done
done

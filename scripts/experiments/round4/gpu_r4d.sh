#!/bin/bash
# round 4, call d: parity after the mark-based cold ranking, RCCL one-device test, A/B vs round 3, conversion timeline
timeout 900 python -m pytest tests/test_gpu_multi.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w r03"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_r03.so one --workload $w
    echo "== $w new"; one --workload $w
  done
done
bash scripts/gpu_convtrace.sh rmat24 62 | tail -34

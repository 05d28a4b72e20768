#!/bin/bash
# round 4, call o: combine whose later rounds load only the runs that are that long (A/B vs the previous library)
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w before"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_cmb0.so one --workload $w
    echo "== $w after"; one --workload $w
  done
done
timeout 600 python -m pytest tests/test_gpu_slabs.py -x -q 2>&1 | tail -2

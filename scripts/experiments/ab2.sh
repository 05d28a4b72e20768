#!/bin/bash
# A/B of two library builds on the bench workloads: scripts/probes/libcsr5hip_prev.so vs the in-tree library
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -2
for rep in 1 2 3; do
  for v in prev cur; do
    if [ $v = prev ]; then export CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_prev.so; else unset CSR5HIP_LIB; fi
    echo -n "$v: "; one
    echo -n "$v: "; one --mode two-pass
    echo -n "$v: "; one --workload webbase --steps 300
    echo -n "$v: "; one --workload nd24k --steps 200
    echo -n "$v: "; one --workload rmat22 --steps 30 --warmup 3
  done
done
unset CSR5HIP_LIB

#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
for rep in 1 2 3; do
  for v in prev cur; do
    if [ $v = prev ]; then export CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_prev.so; else unset CSR5HIP_LIB; fi
    echo -n "$v: "; one --steps 1000
    echo -n "$v: "; one --steps 1000 --sigma 4
  done
done

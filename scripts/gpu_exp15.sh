#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
for v in 4096 8192 2048 4096 8192; do
  echo "== window bytes $v"
  if [ $v != 4096 ]; then export CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_xw$v.so; else unset CSR5HIP_LIB; fi
  one --workload nd24k --steps 100 --sigma 16
  one --workload nd24k --dtype f64 --steps 100 --sigma 16
  one --workload nd24k --dtype f64 --steps 100 --sigma 12
done

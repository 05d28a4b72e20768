#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
for v in base nt base nt; do
  echo "== $v"
  if [ $v = nt ]; then export CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_CSR5_NT_STREAM.so; else unset CSR5HIP_LIB; fi
  one --steps 1000 --sigma 5
  one --workload webbase --steps 300 --sigma 4
  one --workload nd24k --steps 100 --sigma 16
  one --workload rmat22 --steps 30 --warmup 3 --sigma 16
  one --workload rmat20 --steps 100 --warmup 3 --sigma 16
done

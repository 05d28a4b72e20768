#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
for w in 4 2 3 4 2; do
  echo "== waves per workgroup $w"
  if [ $w != 4 ]; then export CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_wpb$w.so; else unset CSR5HIP_LIB; fi
  one --steps 1000 --sigma 4; one --steps 1000 --sigma 5; one --steps 1000 --sigma 8
  one --workload webbase --steps 300 --sigma 4
  one --workload nd24k --steps 100 --sigma 16; one --workload nd24k --steps 100 --sigma 20; one --workload nd24k --steps 100 --sigma 12
  one --workload nd24k --dtype f64 --steps 100 --sigma 16; one --workload nd24k --dtype f64 --steps 100 --sigma 12
  one --workload rmat22 --steps 30 --warmup 3 --sigma 16
done

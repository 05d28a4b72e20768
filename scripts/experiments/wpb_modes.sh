#!/bin/bash
# waves per workgroup x sigma, several processes each (the step time has per-process modes)
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py | awk '{print $(NF-6)}'; }
for s in 5 8; do
  for w in 1 2 4; do
    if [ $w = 2 ]; then unset CSR5HIP_LIB; else export CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_wpb$w.so; fi
    echo -n "sigma=$s wpb=$w: "; for i in 1 2 3 4 5; do one --sigma $s; done | tr '\n' ' '; echo
  done
done
unset CSR5HIP_LIB
for w in 1 2 4; do
  if [ $w = 2 ]; then unset CSR5HIP_LIB; else export CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_wpb$w.so; fi
  echo -n "nd24k wpb=$w: "; for i in 1 2; do one --workload nd24k --steps 200 --sigma 16; done | tr '\n' ' '; echo
  echo -n "webbase wpb=$w: "; for i in 1 2; do one --workload webbase --steps 300 --sigma 8; done | tr '\n' ' '; echo
done

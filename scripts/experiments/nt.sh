#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-30,95-175; }
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
for rep in 1 2; do
for nt in off force; do
  echo -n "nt=$nt "; one --workload rmat22 --steps 30 --warmup 3 --stream-nt $nt
  echo -n "nt=$nt "; one --workload rmat24 --steps 10 --warmup 2 --stream-nt $nt
  echo -n "nt=$nt "; one --workload rmat21 --steps 50 --warmup 3 --stream-nt $nt
done; done

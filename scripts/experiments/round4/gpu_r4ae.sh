#!/bin/bash
# round 4, call ae: slab hash granule (columns hashed together) now that cold gathers read the dense permuted copy, not lines of x
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230; }
for rep in 1 2; do
  for sh in 4 0 2 8; do echo "== rmat24 shift $sh"; one --workload rmat24 --slab-shift $sh; done
  for sh in 4 0 2 8; do echo "== rmat22 shift $sh"; one --workload rmat22 --slab-shift $sh; done
done

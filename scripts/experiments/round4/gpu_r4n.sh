#!/bin/bash
# round 4, call n: slab count with the 16 384-slot table (8 / 16 / 32), R-MAT 24 and 22
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230; }
for rep in 1 2; do
  for s in 16 32 8; do echo "== rmat24 slabs $s"; one --workload rmat24 --slabs $s; done
  for s in 8 16 32; do echo "== rmat22 slabs $s"; one --workload rmat22 --slabs $s; done
done

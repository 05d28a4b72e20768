#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py | awk '{print $(NF-6), $(NF-5)}'; }
for rep in 1 2 3 4 5; do
  echo -n "w50: "; one
  echo -n "w20000: "; one --warmup 20000
  echo -n "w50 steps 20000: "; one --steps 20000
done

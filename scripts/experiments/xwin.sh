#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
for rep in 1 2 3; do
  one
  one --x-window force
  one --lds-y off
  one --x-window force --lds-y off
done

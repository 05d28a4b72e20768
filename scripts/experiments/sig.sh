#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
for s in 4 6 8 12 16; do one --workload webbase --steps 300 --sigma $s; done
for s in 4 5 6 8 12; do one --sigma $s; done
for s in 8 12 16 24; do one --workload rmat22 --steps 30 --warmup 3 --sigma $s; done

#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-30,36-48,95-135; }
for s in 16 20 24 32; do one --workload nd24k --steps 200 --sigma $s; done
for s in 4 6 8; do one --workload webbase --steps 300 --sigma $s; done
for s in 8 16; do one --workload rmat22 --steps 30 --warmup 3 --sigma $s; done
for s in 5 6 8; do one --sigma $s; one --sigma $s; done

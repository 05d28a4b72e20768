#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
for b in 0.3 0.6 0.8 0.95; do one --workload webbase --steps 300 --band $b; done
one --workload webbase --steps 300 --band 0.8 --x-window off
one --workload webbase --steps 300 --band 0.8 --sigma tuned

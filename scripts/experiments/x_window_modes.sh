#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
for xw in off force auto; do
echo "== x-window $xw"
one --workload nd24k --steps 100 --sigma 16 --x-window $xw
one --workload nd24k --dtype f64 --steps 100 --sigma 16 --x-window $xw
one --workload nd24k --steps 300 --scale 0.05 --sigma 16 --x-window $xw
one --workload scircuit --steps 500 --x-window $xw
one --workload scircuit --steps 200 --scale 10 --x-window $xw
one --workload webbase --steps 200 --x-window $xw
done

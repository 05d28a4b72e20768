#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
timeout 900 python -m pytest tests -m gpu -x -q -k "x_window or zoo_integer" 2>&1 | tail -3
one --workload nd24k --steps 100 --sigma 16
one --workload nd24k --steps 100 --sigma 20
one --workload nd24k --steps 100 --sigma 12
one --workload nd24k --dtype f64 --steps 100 --sigma 16
one --workload nd24k --dtype f64 --steps 100 --sigma 12
one --workload nd24k --steps 300 --scale 0.05 --sigma 16

#!/bin/bash
# quick loop: parity tests, then bench lines for the main workloads
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
one --steps 1000; one --steps 1000 --mode two-pass
for s in 4 8 16; do one --steps 1000 --sigma $s; done
one --workload webbase --steps 300
one --workload nd24k --steps 200; one --workload nd24k --steps 200 --lds-y force
one --workload nd24k --dtype f64 --steps 100
one --workload rmat22 --steps 30 --warmup 3

#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
echo "== scircuit sigma sweep (fused)"; for s in 4 5 6 7 8 10 12 16; do one --steps 1000 --sigma $s; done
echo "== xcd remap off"; one --steps 1000 --xcd-remap 0; one --workload nd24k --steps 100 --xcd-remap 0; one --workload webbase --steps 200 --xcd-remap 0; one --workload rmat22 --steps 30 --warmup 3 --xcd-remap 0
echo "== xcd remap on";  one --steps 1000; one --workload nd24k --steps 100; one --workload webbase --steps 200; one --workload rmat22 --steps 30 --warmup 3

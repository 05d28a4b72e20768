#!/bin/bash
# slab structure: parity tests, then bench lines with and without slabs
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_slabs.py -x -q 2>&1 | tail -15
one --workload webbase --steps 300 --slabs 0
one --workload webbase --steps 300
one --workload webbase --steps 300 --slabs 4
one --workload rmat20 --steps 100 --slabs 0
one --workload rmat20 --steps 100
one --workload rmat22 --steps 30 --warmup 3 --slabs 0
for s in 8 16 32; do one --workload rmat22 --steps 30 --warmup 3 --slabs $s; done
one --workload rmat22 --steps 30 --warmup 3 --slabs 32 --slab-shift 0
one --workload rmat24 --steps 10 --warmup 2 --slabs 0
for s in 16 32 64; do one --workload rmat24 --steps 10 --warmup 2 --slabs $s; done

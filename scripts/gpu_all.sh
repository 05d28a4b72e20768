#!/bin/bash
# full GPU suite + a few bench lines
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
one --workload rmat24 --steps 10 --warmup 2 --slabs 8 --slab-hot force
one --workload rmat24 --steps 10 --warmup 2 --slabs 16 --slab-hot force
one --workload rmat22 --steps 30 --warmup 3
one --workload rmat20 --steps 100
one --workload webbase --steps 300

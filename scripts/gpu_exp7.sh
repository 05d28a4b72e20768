#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --workload rmat24 --steps 20 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_rmat24.json | python scripts/benchline.py
python bench.py --workload rmat24 --steps 20 --warmup 3 --no-cpu-baseline --mode two-pass 2>&1 | tail -1 | python scripts/benchline.py

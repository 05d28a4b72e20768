#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4) > gpurun_out/r3x_tests.txt 2>&1
cat gpurun_out/r3x_tests.txt
bash scripts/gpu_convtrace.sh rmat24 52 2>&1 | tee gpurun_out/r3x_convtrace.txt | cut -c1-120 | head -52
for w in scircuit webbase nd24k rmat22; do timeout 300 python scripts/bench_convert.py --workload $w 2>&1 | tail -1; done | tee gpurun_out/r3x_convert.txt

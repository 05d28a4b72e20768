#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
ROUND=r03 bash scripts/gpu_profiles_round.sh > gpurun_out/r3s_profiles.log 2>&1
tail -12 gpurun_out/r3s_profiles.log
(time python bench.py) > gpurun_out/r3s_bench_default.txt 2>&1
tail -c 3000 gpurun_out/r3s_bench_default.txt

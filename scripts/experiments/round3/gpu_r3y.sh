#!/bin/bash
# packed column codes: full GPU suite, A/B against the previous build, conversion trace
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -6) > gpurun_out/r3y_tests.txt 2>&1
cat gpurun_out/r3y_tests.txt
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" scripts/probes/libcsr5hip_prev.so benchmark_spmv_using_csr5_amd/libcsr5hip.so 2>&1 | tee gpurun_out/r3y_ab.txt | cut -c1-200
bash scripts/gpu_convtrace.sh rmat24 60 2>&1 | tee gpurun_out/r3y_convtrace.txt | grep -E "GFLOPS|hot_encode|k_transpose|stats_export|spmv_range|combine" | cut -c1-130

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3f_tests.txt
cat gpurun_out/r3f_tests.txt
bash scripts/experiments/ab_libs.sh "rmat24" benchmark_spmv_using_csr5_amd/libcsr5hip.so scripts/probes/libcsr5hip_cb64.so scripts/probes/libcsr5hip_cb128.so scripts/probes/libcsr5hip_cb512.so scripts/probes/libcsr5hip_cb1024.so scripts/probes/libcsr5hip_cr128.so 2>&1 | tee gpurun_out/r3f_ab.txt

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3e_tests.txt
cat gpurun_out/r3e_tests.txt
{
echo "== rmat24 default"; bash scripts/gpu_kstats.sh --workload rmat24
echo "== rmat24 zero-empty"; bash scripts/gpu_kstats.sh --workload rmat24 --zero-empty 1
echo "== rmat24 32 slabs"; bash scripts/gpu_kstats.sh --workload rmat24 --slabs 32
echo "== rmat22"; bash scripts/gpu_kstats.sh --workload rmat22
echo "== webbase"; bash scripts/gpu_kstats.sh --workload webbase
} 2>&1 | tee gpurun_out/r3e_kstats.txt
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" benchmark_spmv_using_csr5_amd/libcsr5hip.so scripts/probes/libcsr5hip_depth1.so scripts/probes/libcsr5hip_cg1024.so scripts/probes/libcsr5hip_cg4096.so 2>&1 | tee gpurun_out/r3e_ab.txt

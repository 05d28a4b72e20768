#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in pieces2 pieces4; do
  CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so timeout 600 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu -k "hot" 2>&1 | tail -1
done | tee gpurun_out/r3o_tests.txt
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" benchmark_spmv_using_csr5_amd/libcsr5hip.so scripts/probes/libcsr5hip_pieces2.so scripts/probes/libcsr5hip_pieces4.so scripts/probes/libcsr5hip_pieces8.so 2>&1 | tee gpurun_out/r3o_ab.txt

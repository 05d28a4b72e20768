#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in seg256 seg128; do
  CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_parity.py -x -q -m gpu -k "hot or fuzz or rmat" 2>&1 | tail -1
done | tee gpurun_out/r3q_tests.txt
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" benchmark_spmv_using_csr5_amd/libcsr5hip.so scripts/probes/libcsr5hip_seg256.so scripts/probes/libcsr5hip_seg128.so 2>&1 | tee gpurun_out/r3q_ab.txt

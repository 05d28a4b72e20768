#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" benchmark_spmv_using_csr5_amd/libcsr5hip.so scripts/probes/libcsr5hip_abl_ntstore.so 2>&1 | tee gpurun_out/r3k_ab.txt

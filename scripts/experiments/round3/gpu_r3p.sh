#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" benchmark_spmv_using_csr5_amd/libcsr5hip.so scripts/probes/libcsr5hip_abl_gather_nt.so scripts/probes/libcsr5hip_abl_gather_sc1.so scripts/probes/libcsr5hip_abl_gather_sc0sc1.so 2>&1 | tee gpurun_out/r3p_ab.txt

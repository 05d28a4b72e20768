#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "" abl_nocold abl_nogather abl_nostore abl_notable; do
  lib=benchmark_spmv_using_csr5_amd/libcsr5hip.so; [ -n "$v" ] && lib=scripts/probes/libcsr5hip_$v.so
  echo "== ${v:-product}"; CSR5HIP_LIB=$PWD/$lib bash scripts/gpu_kstats.sh --workload rmat24
done 2>&1 | tee gpurun_out/r3h_ablation.txt

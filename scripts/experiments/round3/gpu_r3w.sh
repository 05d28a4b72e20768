#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for v in "" abl_rowscan_noflag abl_rowscan_noempty; do
  lib=benchmark_spmv_using_csr5_amd/libcsr5hip.so; [ -n "$v" ] && lib=scripts/probes/libcsr5hip_$v.so
  echo "== ${v:-product}"; CSR5HIP_LIB=$PWD/$lib bash scripts/gpu_convtrace.sh rmat24 52 2>&1 | grep -E "k_row_scan|k_slab_scatter|k_tile_desc|R-MAT" | cut -c1-110
done 2>&1 | tee gpurun_out/r3w_rowscan.txt
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -2 | tee -a gpurun_out/r3w_rowscan.txt

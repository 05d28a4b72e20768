#!/bin/bash
# round 3, call A: correctness of the range kernel, then same-call A/B against the round-2 library
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r3a_tests.txt
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or full_size or rmat" 2>&1 | tail -15 >> gpurun_out/r3a_tests.txt
cat gpurun_out/r3a_tests.txt
bash scripts/experiments/ab_libs.sh "rmat24 rmat22" scripts/probes/libcsr5hip_prev.so benchmark_spmv_using_csr5_amd/libcsr5hip.so 2>&1 | tee gpurun_out/r3a_ab.txt
for s in 16 32; do
  echo "== slabs $s"; bash scripts/gpu_kstats.sh --workload rmat24 --slabs $s
done 2>&1 | tee gpurun_out/r3a_kstats.txt

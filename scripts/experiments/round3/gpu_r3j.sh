#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -3 > gpurun_out/r3j_tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or full_size or rmat or coupled" 2>&1 | tail -3 >> gpurun_out/r3j_tests.txt
cat gpurun_out/r3j_tests.txt
{
echo "== rmat24 16"; bash scripts/gpu_kstats.sh --workload rmat24
echo "== rmat24 32 slabs"; bash scripts/gpu_kstats.sh --workload rmat24 --slabs 32
echo "== rmat22"; bash scripts/gpu_kstats.sh --workload rmat22
echo "== rmat22 16"; bash scripts/gpu_kstats.sh --workload rmat22 --slabs 16
echo "== webbase"; bash scripts/gpu_kstats.sh --workload webbase
} 2>&1 | tee gpurun_out/r3j_kstats.txt

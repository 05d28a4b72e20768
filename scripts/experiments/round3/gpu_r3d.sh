#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(cd scripts/probes && timeout 300 ./stream_shape) > gpurun_out/r3d_stream_shape.txt 2>&1
cat gpurun_out/r3d_stream_shape.txt
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r3d_tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or full_size or rmat or coupled" 2>&1 | tail -4 >> gpurun_out/r3d_tests.txt
cat gpurun_out/r3d_tests.txt
for s in 16 32; do
  echo "== rmat24 slabs $s"; bash scripts/gpu_kstats.sh --workload rmat24 --slabs $s
done 2>&1 | tee gpurun_out/r3d_kstats.txt
echo "== rmat22"; bash scripts/gpu_kstats.sh --workload rmat22 2>&1 | tee -a gpurun_out/r3d_kstats.txt
echo "== webbase"; bash scripts/gpu_kstats.sh --workload webbase 2>&1 | tee -a gpurun_out/r3d_kstats.txt

#!/bin/bash
# round 3, call B: parallel finish kernel + run-streaming combine: tests, per-kernel times by slab count, all-hot floor
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_slabs.py -x -q -m gpu 2>&1 | tail -6 > gpurun_out/r3b_tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or full_size or rmat or coupled" 2>&1 | tail -6 >> gpurun_out/r3b_tests.txt
cat gpurun_out/r3b_tests.txt
for s in 16 32 64; do
  echo "== rmat24 slabs $s"; bash scripts/gpu_kstats.sh --workload rmat24 --slabs $s
done 2>&1 | tee gpurun_out/r3b_kstats.txt
for s in 8 16; do
  echo "== rmat22 slabs $s"; bash scripts/gpu_kstats.sh --workload rmat22 --slabs $s
done 2>&1 | tee -a gpurun_out/r3b_kstats.txt
echo "== webbase"; bash scripts/gpu_kstats.sh --workload webbase 2>&1 | tee -a gpurun_out/r3b_kstats.txt
timeout 600 python scripts/experiments/hot_floor.py --scale 24 --hubs 131072 2>&1 | tee gpurun_out/r3b_hot_floor.txt

#!/bin/bash
# extra-long fuzz campaign (new seeds): 6 seeds x N small cases, 4 seeds x M cases at 30x the shape
for seed in 101 102 103 104 105 106; do
  CSR5_FUZZ_SEED=$seed CSR5_FUZZ_CASES=${1:-4000} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k seeded_fuzz 2>&1 | tail -1
done
for seed in 201 202 203 204; do
  CSR5_FUZZ_SCALE=30 CSR5_FUZZ_SEED=$seed CSR5_FUZZ_CASES=${2:-800} timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k seeded_fuzz 2>&1 | tail -1
done

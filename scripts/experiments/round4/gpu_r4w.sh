#!/bin/bash
# round 4, call w: fuzz campaign on the final kernels (new seeds) + R-MAT 25 / 26 beyond the BASELINE size
for seed in 401 402 403 404 405; do
  CSR5_FUZZ_SEED=$seed CSR5_FUZZ_CASES=3000 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k seeded_fuzz 2>&1 | tail -1
done
for seed in 501 502 503; do
  CSR5_FUZZ_SCALE=30 CSR5_FUZZ_SEED=$seed CSR5_FUZZ_CASES=600 timeout 1500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k seeded_fuzz 2>&1 | tail -1
done
timeout 900 python scripts/experiments/scale_check.py --scale 25 2>&1 | tail -2
timeout 1200 python scripts/experiments/scale_check.py --scale 26 2>&1 | tail -2

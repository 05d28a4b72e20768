#!/bin/bash
# round 4, call be: long fuzz campaigns on the final sources (narrow-values and x-snapshot modes drawn per case)
for seed in 90001 90002 90003; do
  CSR5_FUZZ_SEED=$seed CSR5_FUZZ_CASES=4000 timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror|assert" | tail -2
done

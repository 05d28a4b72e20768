#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py -x -q -m gpu 2>&1 | tail -4 > gpurun_out/r3n_tests.txt
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "fuzz or full_size or rmat or coupled or checkpoint" 2>&1 | tail -4 >> gpurun_out/r3n_tests.txt
cat gpurun_out/r3n_tests.txt
bash scripts/gpu_convtrace.sh rmat24 52 2>&1 | tee gpurun_out/r3n_convtrace.txt

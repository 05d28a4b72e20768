#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_goldens.py tests/test_gpu_full_size.py tests/test_gpu_slabs.py tests/test_gpu_multi.py -x -q -m gpu 2>&1 | tail -25 > gpurun_out/r3l_tests.txt
timeout 1500 python -m pytest tests/test_bench_contract.py -x -q -m gpu 2>&1 | tail -25 >> gpurun_out/r3l_tests.txt
cat gpurun_out/r3l_tests.txt

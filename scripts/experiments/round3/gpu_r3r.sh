#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(time timeout 3000 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -5) > gpurun_out/r3r_full_gpu_suite.txt 2>&1
cat gpurun_out/r3r_full_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2 | tee gpurun_out/r3r_smoke.txt

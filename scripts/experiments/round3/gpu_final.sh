#!/bin/bash
# last call of the round: full -m gpu suite, smoke, then the round's profiles and the default bench line on the final sources
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
(timeout 2400 python -m pytest tests/ -x -q -m gpu 2>&1 | tail -4) > gpurun_out/final_tests.txt 2>&1
cat gpurun_out/final_tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1 | tee gpurun_out/final_smoke.txt
bash scripts/experiments/round3/gpu_r3s.sh

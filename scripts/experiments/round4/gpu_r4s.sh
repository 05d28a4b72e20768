#!/bin/bash
# round 4, call s: final row-block table (all 15 lines, same box) + the round's profiles
bash scripts/experiments/round4/gpu_r4r.sh > /dev/null 2>&1
cat gpurun_out/r04_shards.txt | cut -c1-120 | tail -32
ROUND=r04 timeout 2400 bash scripts/gpu_profiles_round.sh 2>&1 | tail -12

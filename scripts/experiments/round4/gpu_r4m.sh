#!/bin/bash
# round 4, call m: parity with 8 wavefronts per CU, blocks of R-MAT 24 alone, small configs unchanged?
timeout 1200 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py tests/test_gpu_multi.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
timeout 600 python scripts/experiments/shard_alone.py --scale 24 --world 1 --ranks 0
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7 --slabs 8
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 4 --ranks 0,3
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 2 --ranks 0,1

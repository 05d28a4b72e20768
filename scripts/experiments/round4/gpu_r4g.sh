#!/bin/bash
# round 4, call g: row blocks of R-MAT 24 alone (8 blocks: ranks 0 3 7; slabs auto vs 16), whole matrix first
timeout 600 python scripts/experiments/shard_alone.py --scale 24 --world 1 --ranks 0
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7 --slabs 16

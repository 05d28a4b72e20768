#!/bin/bash
# round 4, call r: all row blocks of R-MAT 24 alone, row weight 2 (default) and 3; whole matrix; R-MAT 22 / 20 sanity
mkdir -p gpurun_out
{
echo "## scripts/experiments/shard_alone.py: every row block of the strong-scaling R-MAT 24 ALONE on one MI355X (cost balance nnz + 2*rows), round-4 kernels, x snapshot"
timeout 600 python scripts/experiments/shard_alone.py --scale 24 --world 1 --ranks 0
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 2 --ranks 0,1
timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 4 --ranks 0,1,2,3
timeout 1200 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,1,2,3,4,5,6,7
echo "## row weight 3"
timeout 1200 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,1,2,3,4,5,6,7 --row-weight 3
echo "## row weight 1"
timeout 1200 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7 --row-weight 1
} 2>/dev/null | tee gpurun_out/r04_shards.txt

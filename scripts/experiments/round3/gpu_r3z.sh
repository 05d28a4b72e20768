#!/bin/bash
# every row block of the strong-scaling R-MAT 24 ALONE on the one GPU, round-3 kernels (estimate of the N-GPU step: the slowest block)
cd ${GRAFT_REPO_ROOT:-/root/repo}
{
echo "## scripts/experiments/shard_alone.py: every row block of the strong-scaling R-MAT 24 ALONE on one MI355X (cost balance nnz + 2*rows), round-3 kernels"
python scripts/experiments/shard_alone.py --scale 24 --world 1 --ranks 0
python scripts/experiments/shard_alone.py --scale 24 --world 2 --ranks 0,1
python scripts/experiments/shard_alone.py --scale 24 --world 4 --ranks 0,1,2,3
python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,1,2,3,4,5,6,7
} 2>&1 | grep -v "amdgpu.ids" | tee gpurun_out/r3z_shards.txt

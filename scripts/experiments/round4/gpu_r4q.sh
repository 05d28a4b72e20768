#!/bin/bash
# round 4, call q: how often must a column be used to earn a table slot?  (blocks of R-MAT 24 alone: their columns are used 1/8 as often)
for u in 48 16 4; do
  echo "== min uses $u, auto slabs"; CSR5_EXPERIMENT_HOT_MIN_USES=$u timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7
  echo "== min uses $u, 16 slabs"; CSR5_EXPERIMENT_HOT_MIN_USES=$u timeout 900 python scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 0,3,7 --slabs 16
done
echo "== whole matrix, min uses 48 / 4"; CSR5_EXPERIMENT_HOT_MIN_USES=48 timeout 600 python scripts/experiments/shard_alone.py --scale 24 --world 1 --ranks 0
CSR5_EXPERIMENT_HOT_MIN_USES=4 timeout 600 python scripts/experiments/shard_alone.py --scale 24 --world 1 --ranks 0

#!/bin/bash
# round 4, call aw: per-kernel times of row blocks 1 and 3 of 8 (R-MAT 24) alone on the GPU: where do the blocks' 170-178 us go?
cd /tmp && export TMPDIR=/tmp
for r in 1 3 7; do
  rm -rf /tmp/pc; timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks $r 2>/dev/null | grep '"rank"' | cut -c1-170
  grep -h "k_spmv_range\|k_slab_combine\|k_range_finish\|k_x_permute" $(find /tmp/pc -name "*kernel_stats.csv") | sed 's/^"void csr5::\([a-z_]*\).*)",/\1 /' | cut -c1-90
done

#!/bin/bash
# round 4, call as: TIMING of k_slab_combine if a row block's S runs of partials lay one behind the other in P (block-major P instead of
# slab-major; results wrong by design: only the addresses change) -- is the combine held by its 16 scattered short runs per block?
cd /tmp && export TMPDIR=/tmp
for m in 0 2 0 2; do
  rm -rf /tmp/pc; CSR5_COMBINE_ATOMIC=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 > /dev/null 2>&1
  echo "mode $m:"; grep -h "combine" $(find /tmp/pc -name "*kernel_stats.csv") | sed 's/.*)",//' | cut -c1-80
done
for m in 0 2; do
  rm -rf /tmp/pc; CSR5_COMBINE_ATOMIC=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --workload rmat22 --steps 20 --warmup 5 > /dev/null 2>&1
  echo "rmat22 mode $m:"; grep -h "combine" $(find /tmp/pc -name "*kernel_stats.csv") | sed 's/.*)",//' | cut -c1-80
done

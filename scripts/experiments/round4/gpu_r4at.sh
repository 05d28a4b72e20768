#!/bin/bash
# round 4, call at: ablations of k_slab_combine on R-MAT 24 (wrong results by design): 3 = no partial / row-byte loads, 6 = no partial
# loads, 7 = no row-byte loads, 4 = no stores, 5 = no LDS accumulation, 0 = product
cd /tmp && export TMPDIR=/tmp
for m in 0 3 6 7 4 5 0; do
  rm -rf /tmp/pc; CSR5_COMBINE_ATOMIC=$m timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 > /dev/null 2>&1
  echo -n "mode $m: "; grep -h "combine" $(find /tmp/pc -name "*kernel_stats.csv") | sed 's/.*)",//' | cut -c1-80
done

#!/bin/bash
# round 4, call ax: gaps between the three kernels of a step inside the replayed graph (row block 3 of 8, and the whole matrix)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pc; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks 3 2>/dev/null | grep '"rank"' | cut -c1-150
python $GRAFT_REPO_ROOT/scripts/experiments/kernel_gaps.py $(find /tmp/pc -name "*kernel_trace.csv") | grep "range\|combine\|finish"
rm -rf /tmp/pc; timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 2>/dev/null | tail -1 | python $GRAFT_REPO_ROOT/scripts/benchline.py | cut -c60-140
python $GRAFT_REPO_ROOT/scripts/experiments/kernel_gaps.py $(find /tmp/pc -name "*kernel_trace.csv") | grep "range\|combine\|finish"

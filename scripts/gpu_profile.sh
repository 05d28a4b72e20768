#!/bin/bash
# rocprofv3 kernel-trace summaries for the bench workloads; outputs under gpurun_out/prof_<tag>/
set -x
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
prof() { # tag, bench args...
  tag=$1; shift
  rocprofv3 --kernel-trace --stats -d $OUT/prof_$tag -o $tag -- python $REPO/bench.py --no-cpu-baseline "$@" > $OUT/prof_$tag.log 2>&1
  tail -1 $OUT/prof_$tag.log
  f=$(find $OUT/prof_$tag -name "*kernel_stats.csv" | head -1)
  [ -n "$f" ] && head -8 "$f"
}
prof scircuit_fused --mode fused --steps 1000
prof scircuit_twopass --mode two-pass --steps 1000
prof scircuit_fused_s16 --mode fused --steps 1000 --sigma 16
prof webbase_fused --workload webbase --mode fused --steps 300
prof webbase_twopass --workload webbase --mode two-pass --steps 300
prof rmat22_twopass --workload rmat22 --mode two-pass --steps 50 --warmup 5
ls -R $OUT | head -50

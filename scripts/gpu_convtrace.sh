#!/bin/bash
# timeline of the last asCSR5 of a bench workload (rocprofv3 kernel trace).  usage: gpu_convtrace.sh <workload> [kernels]
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ct && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ct -o t -- python $REPO/bench.py --no-cpu-baseline --no-sub-configs --workload $1 --steps 3 --warmup 1 > /tmp/ct.log 2>&1
grep '"metric"' /tmp/ct.log | tail -1 | python $REPO/scripts/benchline.py
f=$(find /tmp/ct -name "*kernel_trace.csv" | head -1)
python $REPO/scripts/experiments/conv_trace.py $f ${2:-60}

#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
for w in nd24k scircuit; do
  for rk in off force; do
    echo "== $w range-kernel $rk"
    timeout 300 python bench.py --workload $w --no-cpu-baseline --range-kernel $rk 2>&1 | grep '^{' | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('  range_kernel', d['config']['range_kernel'], 'cold us', r['launch_us'], 'frac', r['frac'], '| warm us', r['warm']['launch_us'], 'frac', r['warm']['frac'])"
  done
done 2>&1 | tee gpurun_out/r3t_range_plain.txt

#!/bin/bash
# round 4, call e: sigma sweep on the round-4 kernels (fp32 table), cold knobs of the small configs
mkdir -p gpurun_out
cold() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('   cold %.2f us frac %.3f | warm %.2f us frac %.3f | sigma %d' % (r['launch_us'], r['frac'], r['warm']['launch_us'], r['warm']['frac'], d['config']['sigma']))"; }
for w in nd24k; do
  echo "== $w default"; cold --workload $w
  echo "== $w nt force"; cold --workload $w --stream-nt force
  echo "== $w sigma 32"; cold --workload $w --sigma 32
  echo "== $w sigma 24"; cold --workload $w --sigma 24
  echo "== $w sigma 12"; cold --workload $w --sigma 12
  echo "== $w lds-y force"; cold --workload $w --lds-y force
done
for w in scircuit webbase; do
  echo "== $w default"; cold --workload $w
  echo "== $w nt force"; cold --workload $w --stream-nt force
  echo "== $w sigma 8"; cold --workload $w --sigma 8
  echo "== $w sigma 16"; cold --workload $w --sigma 16
done
timeout 1500 python scripts/experiments/sigma_table.py > gpurun_out/r04_sigma_table.txt 2>&1; tail -3 gpurun_out/r04_sigma_table.txt

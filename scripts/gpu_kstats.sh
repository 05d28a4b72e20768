#!/bin/bash
# usage: gpu_kstats.sh <bench args...>  -> average duration of the step kernels under rocprofv3 --kernel-trace --stats
REPO=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -o t -- python $REPO/bench.py --no-cpu-baseline --no-sub-configs "$@" > /tmp/ks.log 2>&1
grep '"metric"' /tmp/ks.log | tail -1 | python $REPO/scripts/benchline.py
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/ks/**/*kernel_stats.csv', recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r['Name'] for k in ('k_spmv', 'k_slab_combine', 'k_calibrate', 'k_range')):
        print(f"   {r['Name'].split('(')[0][:70]:70s} calls={r['Calls']:>5s} avg={float(r['AverageNs'])/1e3:9.2f} us")
PY

#!/bin/bash
# round 4, call f: parity after the k_range_finish rewrite; fp64 sigma 10 vs 16 on nd24k-like fp64 and R-MAT
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
cold() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python -c "
import json,sys
d=json.loads(sys.stdin.readline()); r=d['roofline']; print('   cold %.2f us frac %.3f | warm %.2f us frac %.3f | sigma %d xwin %s' % (r['launch_us'], r['frac'], r['warm']['launch_us'], r['warm']['frac'], d['config']['sigma'], d['config']['lds_x_window']))"; }
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230; }
for s in -1 10 12 8; do echo "== nd24k f64 sigma $s"; cold --workload nd24k --dtype f64 --sigma $s; done
for s in -1 10; do echo "== rmat22 sigma $s"; one --workload rmat22 --sigma $s; echo "== rmat24 sigma $s"; one --workload rmat24 --sigma $s; done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d /tmp/ks -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 3 > /dev/null 2>&1; f=$(find /tmp/ks -name "*kernel_stats.csv" | head -1); grep -E "k_spmv_range|k_range_finish|k_slab_combine|k_x_permute" $f | cut -c1-200

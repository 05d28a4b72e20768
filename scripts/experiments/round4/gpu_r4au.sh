#!/bin/bash
# round 4, call au: the combine as a stream, second form (unit of work = one round of a block, next round's loads issued first, ds_add_f64)
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22; do for m in 0 1 0 1; do echo -n "$w stream=$m: "; CSR5_COMBINE_STREAM=$m one --workload $w; done; done
CSR5_COMBINE_STREAM=1 timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
cd /tmp && export TMPDIR=/tmp
CSR5_COMBINE_STREAM=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 > /dev/null 2>&1
grep -h "combine" $(find /tmp/pc -name "*kernel_stats.csv") | sed 's/.*)",//' | cut -c1-80

#!/bin/bash
# round 4, call ao: the combine as a stream (persistent wavefronts, two row blocks in flight each); CSR5_COMBINE_STREAM = workgroups per CU
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22; do for m in 0 3 2 4 0 3; do echo -n "$w stream=$m: "; CSR5_COMBINE_STREAM=$m one --workload $w; done; done
CSR5_COMBINE_STREAM=3 timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
cd /tmp && export TMPDIR=/tmp
CSR5_COMBINE_STREAM=3 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 > /dev/null 2>&1
grep -h "combine\|k_spmv_range\|range_finish" $(find /tmp/pc -name "*kernel_stats.csv") | cut -c1-150

#!/bin/bash
# Round profiles: rocprofv3 kernel-trace stats + PMC FETCH/WRITE passes for the bench workloads.
# Output: gpurun_out/profiles_<tag>/...  (copy the summaries you want judged into profiles/)
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # tag, bench args
  tag=$1; shift
  d=$OUT/profiles_$tag; mkdir -p $d
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -o t -- python $REPO/bench.py --no-cpu-baseline "$@" > $d/bench_under_rocprof.log 2>&1
  grep '"metric"' $d/bench_under_rocprof.log | tail -1 > $d/bench_line.json
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | tr ' ' '_')
    timeout 300 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d/pmc_$n -o p -- python $REPO/bench.py --no-cpu-baseline --steps 20 --warmup 5 "${@:1}" > $d/pmc_$n.log 2>&1
  done
  find $d -name "*.csv" | head -20
}
run scircuit_fused
run scircuit_twopass --mode two-pass
run nd24k_fused --workload nd24k --steps 200
run webbase_fused --workload webbase --steps 300
run rmat22_fused --workload rmat22 --steps 50 --warmup 5
# ingest: kernel-trace stats of file -> CSR in HBM, general and symmetric
for kind in general symmetric; do
  d=$OUT/profiles_ingest_$kind; mkdir -p $d
  extra=""; [ $kind = symmetric ] && extra="--symmetric"
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -o t -- python $REPO/scripts/bench_ingest.py --entries 10000000 --repeat 2 $extra > $d/bench_under_rocprof.log 2>&1
  grep '"metric"' $d/bench_under_rocprof.log | tail -1 > $d/bench_line.json
done

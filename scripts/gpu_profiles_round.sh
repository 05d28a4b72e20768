#!/bin/bash
# Profiles of a round (ROUND=r03 by default): per workload, rocprofv3 kernel-trace stats of `python bench.py` + separate PMC passes (FETCH_SIZE,
# WRITE_SIZE, TCC hit/miss; counters only), parsed ON THE BOX into small files under gpurun_out/profiles_$ROUND/
# (the raw rocprofv3 directories are deleted: gpurun copies back at most 64 MiB).
REPO=${GRAFT_REPO_ROOT:-/root/repo}
ROUND=${ROUND:-r06}
OUT=$REPO/gpurun_out/profiles_$ROUND
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
run() { # tag, bench args
  tag=$1; shift
  d=/tmp/prof_$tag; rm -rf $d; mkdir -p $d
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -o t -- python $REPO/bench.py --no-cpu-baseline --no-sub-configs --no-side-figures "$@" > $d/bench_under_rocprof.log 2>&1
  grep '"metric"' $d/bench_under_rocprof.log | tail -1 > $OUT/${tag}_bench_line.json
  cp $(find $d/trace -name "*kernel_stats.csv" | head -1) $OUT/${tag}_kernel_stats.csv 2>/dev/null
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    n=$(echo $c | tr ' ' '_')
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $d/pmc_$n -o p -- python $REPO/bench.py --no-cpu-baseline --no-sub-configs --no-side-figures --steps 10 --warmup 2 "$@" > $d/pmc_$n.log 2>&1
  done
  python3 $REPO/scripts/profile_parse.py $tag $d $OUT
  rm -rf $d
}
# RUNS = which of the runs below to make (default: all)
RUNS=${RUNS:-"rmat24 rmat24_noslab rmat22 webbase webbase_noslab scircuit nd24k"}
want() { case " $RUNS " in *" $1 "*) return 0;; esac; return 1; }
want rmat24 && run rmat24 --steps 20 --warmup 5
want rmat24_noslab && run rmat24_noslab --steps 10 --warmup 2 --slabs 0
want rmat22 && run rmat22 --workload rmat22 --steps 30 --warmup 5
want webbase && run webbase --workload webbase --steps 300 --no-cold
want webbase_noslab && run webbase_noslab --workload webbase --steps 300 --slabs 0 --no-cold
want scircuit && run scircuit --workload scircuit --steps 1000 --no-cold
want nd24k && run nd24k --workload nd24k --steps 200 --no-cold
ls -la $OUT

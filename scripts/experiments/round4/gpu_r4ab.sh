#!/bin/bash
# round 4, call ab: where does the final tile kernel's time go?  builds of the product with one piece of work removed
# (scripts/experiments/hot_ablation.py; wrong results by design), kernel averages from rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
for v in product nocold nogather nostore notable; do
  lib=$GRAFT_REPO_ROOT/scripts/probes/libcsr5hip_abl_$v.so
  [ $v = product ] && lib=$GRAFT_REPO_ROOT/benchmark_spmv_using_csr5_amd/libcsr5hip.so
  rm -rf /tmp/ks_$v
  CSR5HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks_$v -o k -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 3 > /tmp/ks_$v.log 2>&1
  f=$(find /tmp/ks_$v -name "*kernel_stats.csv" | head -1)
  echo "== $v: $(grep '"metric"' /tmp/ks_$v.log | tail -1 | python $GRAFT_REPO_ROOT/scripts/benchline.py | cut -c88-130)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r["Name"] for k in ("k_spmv_range", "k_slab_combine", "k_range_finish")):
        print("   %-40s calls %3s avg %8.1f us" % (r["Name"].split("(")[0][5:45], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done

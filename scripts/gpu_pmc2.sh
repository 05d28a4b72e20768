#!/bin/bash
# instruction-fetch / scalar-cache counters for one bench workload
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_IFETCH_LEVEL" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_SMEM SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES SQ_WAIT_INST_LDS SQ_WAVE_CYCLES" ; do
  i=$((i+1))
  rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc2_${tag}_$i -o p -- python $REPO/bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > $OUT/pmc2_${tag}_$i.log 2>&1
  f=$(find $OUT/pmc2_${tag}_$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $grp"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if "k_spmv" in k or "k_calibrate" in k:
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print("  ", k)
    for c, v in cs.items():
        print("      %-32s n=%3d avg=%.1f" % (c, len(v), sum(v) / len(v)))
PY
done

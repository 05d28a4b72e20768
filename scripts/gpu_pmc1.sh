#!/bin/bash
# ONE rocprofv3 PMC pass (counters in $PMC, space separated; kernels whose name contains $KFILTER, default k_spmv) for a bench workload.  usage: PMC="A B" gpu_pmc1.sh <tag> <bench args...>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
tag=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --pmc $PMC --kernel-trace --output-format csv -d $OUT/pmc1_${tag} -o p -- python $REPO/bench.py --no-cpu-baseline --steps 20 --warmup 5 "$@" > $OUT/pmc1_${tag}.log 2>&1
f=$(find $OUT/pmc1_${tag} -name "*counter_collection.csv" | head -1)
echo "== $tag: $PMC"
[ -n "$f" ] && KFILTER=${KFILTER:-k_spmv} python3 - "$f" <<'PY'
import csv, sys, collections, os
agg = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if os.environ.get("KFILTER", "k_spmv") in r.get("Kernel_Name", ""):
        agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
last = int(os.environ.get("KLAST", "0"))  # > 0: only the last KLAST dispatches (the cold-protocol launches of a bench run come last)
for c, v in sorted(agg.items()):
    v = v[-last:] if last > 0 else v
    print(f"   {c:34s} n={len(v):3d} avg={sum(v)/len(v):14.1f}")
PY

#!/bin/bash
# rocprofv3 PMC passes (counters only, one group per pass) for a bench workload.
# usage: gpu_pmc.sh <tag> <bench args...>
REPO=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$REPO/gpurun_out
tag=$1; shift
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
GROUPS_DEFAULT=1
if [ -n "$PMC_GROUPS" ]; then GROUPS_DEFAULT=0; fi
for grp in ${PMC_GROUPS:+"$PMC_GROUPS"} "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" \
           "FETCH_SIZE" "WRITE_SIZE" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" ; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $grp --kernel-trace --output-format csv -d $OUT/pmc_${tag}_$i -o p -- python $REPO/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 "$@" > $OUT/pmc_${tag}_$i.log 2>&1
  f=$(find $OUT/pmc_${tag}_$i -name "*counter_collection.csv" | head -1)
  echo "== pass $i: $grp -> $f"
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    k = r.get("Kernel_Name", "")
    if "k_spmv" in k or "k_calibrate" in k or "k_slab" in k:
        agg[k[:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in agg.items():
    print("  ", k)
    for c, v in cs.items():
        print("      %-32s n=%3d avg=%.1f" % (c, len(v), sum(v) / len(v)))
PY
  rm -rf $OUT/pmc_${tag}_$i   # the parsed averages above are what is kept (gpurun copies back at most 64 MiB)
done

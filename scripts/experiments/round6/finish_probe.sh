#!/bin/bash
# k_range_finish of one row block of R-MAT 24 under rocprofv3 (block 1 of 8: hub rows), for every library given as
# CSR5HIP_LIB-style path in "$@" ("product" = the in-tree library).  Round 6: timing-only ablations built by patching a
# temporary copy of csr5_hot.hip (no bisection of `head` / no tail-row loop / no P[row] load) located the kernel's extra 7 us
# on hub blocks in the serial LDS walk of long tail rows (profiles/r06_probes.txt section 5).
cd /tmp && export TMPDIR=/tmp
for v in product "$@"; do
  lib=""; [ "$v" != product ] && lib="$v"
  rm -rf /tmp/sa
  CSR5HIP_LIB=${lib:-$GRAFT_REPO_ROOT/benchmark_spmv_using_csr5_amd/libcsr5hip.so} rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sa -o t -- \
    python $GRAFT_REPO_ROOT/scripts/experiments/shard_alone.py --scale 24 --world 8 --ranks ${RANK:-1} --x-snapshot 1 --steps 30 > /tmp/sa.log 2>&1
  python - "$v" <<PY
import csv,glob,sys
f=glob.glob("/tmp/sa/**/*kernel_stats.csv",recursive=True)[0]
for r in csv.DictReader(open(f)):
    if any(k in r["Name"] for k in ("k_range_finish", "k_spmv_range", "k_slab_combine")):
        print("%-40s %-44s avg %7.2f us min %7.2f" % (sys.argv[1][-40:], r["Name"].split("(")[0][-44:], float(r["AverageNs"])/1e3, float(r["MinNs"])/1e3))
PY
done

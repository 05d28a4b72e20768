#!/bin/bash
# cold-protocol figures of the small configs under each path (which path should the auto rule pick when the working set is cold?)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
out=gpurun_out/r3_coldpaths.txt
: > $out
run() { echo "## $*" >> $out; timeout 300 python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>/dev/null | python -c '
import sys, json
for line in sys.stdin:
    line=line.strip()
    if not line.startswith("{"): continue
    d=json.loads(line); r=d.get("roofline",{})
    print("   cold_us=%s frac=%s warm=%s slabs=%s" % (r.get("launch_us"), r.get("frac"), (r.get("warm") or {}).get("launch_us"), d.get("config",{}).get("column_slabs")))
' >> $out; }
for w in webbase scircuit; do
  run --workload $w
  run --workload $w --slabs 0
  run --workload $w --slabs 0 --x-window off
  run --workload $w --slabs 2
  run --workload $w --slabs 8
  run --workload $w --mode two-pass --slabs 0
done
run --workload nd24k --dtype f32
run --workload nd24k --dtype f32 --x-window off
run --workload nd24k --dtype f32 --sigma 8
run --workload nd24k --dtype f32 --sigma 32
cat $out

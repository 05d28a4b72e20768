#!/bin/bash
which numactl; lscpu | grep -i "numa\|socket\|model name" | head -8
cat /sys/class/drm/card*/device/numa_node 2>/dev/null | head -4
one() { "$@" python bench.py --no-cpu-baseline --spinup-seconds 0.5 2>&1 | tail -1 | python scripts/benchline.py | cut -c100-140; }
N=$(nproc)
H=$((N/2))
for rep in 1 2 3; do
  echo -n "cpus 0-7:        "; one taskset -c 0-7
  echo -n "cpus $((H/2))-$((H/2+7)):    "; one taskset -c $((H/2))-$((H/2+7))
  echo -n "cpus $H-$((H+7)):  "; one taskset -c $H-$((H+7))
  echo -n "cpus $((N-8))-$((N-1)): "; one taskset -c $((N-8))-$((N-1))
done

#!/bin/bash
# round 4, call ba: did the storage-type template parameter change the plain fp64 kernel?  previous commit vs current, same call
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22; do for v in prev cur prev cur; do echo -n "$w $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload $w; done; done

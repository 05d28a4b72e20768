#!/bin/bash
# round 4, call bh: the combine's loads of P (and of the row bytes) with the non-temporal hint -- P is dead after the combine
one() { python bench.py --no-cpu-baseline --no-sub-configs --no-side-figures "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22 webbase; do for v in base abl_combine_nt abl_combine_ntP base abl_combine_nt abl_combine_ntP; do echo -n "$w $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload $w --no-cold; done; done

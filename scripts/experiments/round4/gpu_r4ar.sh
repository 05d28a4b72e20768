#!/bin/bash
# round 4, call ar: workgroup size of k_slab_combine (one row block per wavefront): 64 / 128 / 256 (product) / 512 / 1024 threads
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22 webbase; do for b in 256 64 128 512 1024 256; do
  lib=$PWD/scripts/probes/libcsr5hip_cb$b.so; [ $b = 256 ] && lib=$PWD/benchmark_spmv_using_csr5_amd/libcsr5hip.so
  echo -n "$w combine block $b: "; CSR5HIP_LIB=$lib one --workload $w --no-cold; done; done

#!/bin/bash
# round 4, call p: windowed y-compaction: region 4096 B (whole tile, table 16 384 slots) / 2048 / 1024 / 512 B per wavefront
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|rror" | tail -5
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-200; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w region 4096"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_wl4096.so one --workload $w
    echo "== $w region 2048"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_wl2048.so one --workload $w
    echo "== $w region 1024"; one --workload $w
    echo "== $w region 512"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_wl512.so one --workload $w
  done
done

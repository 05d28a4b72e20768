#!/bin/bash
# round 4, call a: parity of the permuted-x path, then same-call A/B against the round-3 library
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_goldens.py tests/test_gpu_parity.py -x -q 2>&1 | tail -8
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230; }
for rep in 1 2; do
  for w in rmat24 rmat22; do
    echo "== $w r03"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_r03.so one --workload $w
    echo "== $w new snapshot"; one --workload $w
    echo "== $w new live"; one --workload $w --x-snapshot 0
  done
done
python bench.py --no-cpu-baseline --no-sub-configs 2>&1 | tail -1 > gpurun_out/r4a_rmat24.json

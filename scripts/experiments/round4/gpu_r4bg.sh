#!/bin/bash
# round 4, call bg: y_offset prefix of the range kernel by six DPP adds instead of six __shfl_up (ds_bpermute) steps; parity, then same-call pairs
timeout 900 python -m pytest tests/test_gpu_slabs.py tests/test_gpu_full_size.py -x -q 2>&1 | grep -E "passed|failed|rror|assert" | tail -3
CSR5_FUZZ_SEED=515 CSR5_FUZZ_CASES=1500 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
one() { python bench.py --no-cpu-baseline --no-sub-configs --no-side-figures "$@" 2>&1 | tail -1 | python scripts/benchline.py | cut -c60-140; }
for w in rmat24 rmat22; do for v in base dppscan base dppscan base dppscan; do echo -n "$w $v: "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_$v.so one --workload $w; done; done

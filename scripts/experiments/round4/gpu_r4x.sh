#!/bin/bash
# round 4, call x: does a denser column sample (1 chunk in 16 / 8 instead of 64) order the cold regions better?  traffic + time + conversion
for cap in 64 16 8; do
  echo "== stride cap $cap"
  export CSR5_EXPERIMENT_STRIDE_CAP=$cap
  python bench.py --no-cpu-baseline --no-sub-configs 2>&1 | tail -1 | python scripts/benchline.py | cut -c1-230
  PMC="FETCH_SIZE" KFILTER=k_ bash scripts/gpu_pmc1.sh f$cap --no-sub-configs | grep -v "^$"
  PMC="WRITE_SIZE" KFILTER=k_ bash scripts/gpu_pmc1.sh w$cap --no-sub-configs | grep -v "^$"
  PMC="TCC_MISS_sum TCC_HIT_sum" KFILTER=k_spmv_range bash scripts/gpu_pmc1.sh m$cap --no-sub-configs | grep -v "^$"
  rm -rf gpurun_out/pmc1_*
done

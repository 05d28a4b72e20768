#!/bin/bash
# round 4, call av: k_slab_combine with 32-bit byte offsets on a scalar base (288 -> 256 vector instructions per block)
cd /tmp && export TMPDIR=/tmp
for v in product off32 product off32; do
  lib=$GRAFT_REPO_ROOT/scripts/probes/libcsr5hip_$v.so; [ $v = product ] && lib=$GRAFT_REPO_ROOT/benchmark_spmv_using_csr5_amd/libcsr5hip.so
  rm -rf /tmp/pc; CSR5HIP_LIB=$lib timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pc -o t -- python $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-sub-configs --steps 20 --warmup 5 > /dev/null 2>&1
  echo -n "$v: "; grep -h "combine" $(find /tmp/pc -name "*kernel_stats.csv") | sed 's/.*)",//' | cut -c1-80
done

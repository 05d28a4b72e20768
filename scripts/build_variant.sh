#!/bin/bash
# Experiment build of libcsr5hip with extra defines, for same-call A/B runs (scripts/experiments/ab_libs.sh).
#   scripts/build_variant.sh <name> "<-D flags>"   ->  scripts/probes/libcsr5hip_<name>.so   (git-ignored; travels with gpurun)
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/benchmark_spmv_using_csr5_amd/csrc
out=/tmp/csr5_variant_$name
mkdir -p $out
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -I$root/include -I$src -DCSR5_FEW_SIGMAS $flags"
pids=()
for f in csr5_format csr5_capi csr5_ingest csr5_slab csr5_multi csr5_hot; do
  /opt/rocm/bin/hipcc $HIPFLAGS -c $src/$f.hip -o $out/$f.o & pids+=($!)
done
/opt/rocm/bin/hipcc $HIPFLAGS -DCSR5_SPMV_ONLY_F64 -c $src/csr5_spmv.hip -o $out/csr5_spmv_f64.o & pids+=($!)
/opt/rocm/bin/hipcc $HIPFLAGS -DCSR5_SPMV_ONLY_F32 -c $src/csr5_spmv.hip -o $out/csr5_spmv_f32.o & pids+=($!)
for p in "${pids[@]}"; do wait $p; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $root/scripts/probes/libcsr5hip_$name.so $out/*.o -ldl
ls -la $root/scripts/probes/libcsr5hip_$name.so

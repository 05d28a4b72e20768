#!/bin/bash
# Experiment build of libcsr5hip that differs from the product build only in csr5_walk.hip's defines (same-call A/B runs through
# CSR5HIP_LIB):   scripts/build_walk_variant.sh <name> "<-D flags>"   ->  scripts/probes/libcsr5hip_<name>.so (git-ignored; travels with gpurun)
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/benchmark_spmv_using_csr5_amd/csrc
out=/tmp/csr5_walkvar_$name
mkdir -p $out
make -C $src -j8 all > /dev/null
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -I$root/include -I$src $flags"
/opt/rocm/bin/hipcc $HIPFLAGS -DCSR5_WALK_ONLY_F64 -c $src/csr5_walk.hip -o $out/csr5_walk_f64.o &
/opt/rocm/bin/hipcc $HIPFLAGS -DCSR5_WALK_ONLY_F32 -c $src/csr5_walk.hip -o $out/csr5_walk_f32.o &
wait
others=$(ls $src/build/*.o | grep -v csr5_walk)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $root/scripts/probes/libcsr5hip_$name.so $others $out/*.o -ldl
ls -la $root/scripts/probes/libcsr5hip_$name.so

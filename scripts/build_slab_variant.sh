#!/bin/bash
# Experiment build of libcsr5hip that differs from the product build only in csr5_slab.hip's defines (CSR5HIP_LIB A/B runs):
#   scripts/build_slab_variant.sh <name> "<-D flags>"   ->  scripts/probes/libcsr5hip_<name>.so (git-ignored; travels with gpurun)
set -e
name=$1; flags=$2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/benchmark_spmv_using_csr5_amd/csrc
out=/tmp/csr5_slabvar_$name
mkdir -p $out
make -C $src -j8 all > /dev/null
HIPFLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wall -Wno-unused-function -I$root/include -I$src $flags"
/opt/rocm/bin/hipcc $HIPFLAGS -c $src/csr5_slab.hip -o $out/csr5_slab.o
others=$(ls $src/build/*.o | grep -v csr5_slab.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o $root/scripts/probes/libcsr5hip_$name.so $others $out/*.o -ldl
ls -la $root/scripts/probes/libcsr5hip_$name.so

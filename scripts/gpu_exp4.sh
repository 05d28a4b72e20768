#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['workload'][:14], d['dtype'],'sigma',d['config']['sigma'],'GFLOPS',d['value'],'us',d['roofline']['launch_us'],'GB/s',d['roofline']['achieved'])"; }
echo "== nd24k f32 sigma sweep"; for s in 8 12 16 20 24 32; do one --workload nd24k --steps 100 --sigma $s; done
echo "== nd24k f64"; for s in 8 16 32; do one --workload nd24k --dtype f64 --steps 100 --sigma $s; done
echo "== nd24k f32 ablations (1 = no gather, 3 = no gather no interior stores)"
for a in 1 3; do for s in 16 32; do CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_ablate$a.so one --workload nd24k --steps 100 --sigma $s; done; done
echo "== rmat22 sigma sweep"; for s in 4 8 16 32; do one --workload rmat22 --steps 30 --warmup 3 --sigma $s; done
echo "== rmat22 no gather"; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_ablate1.so one --workload rmat22 --steps 30 --warmup 3 --sigma 16
echo "== webbase sigma sweep"; for s in 4 5 8 16; do one --workload webbase --steps 200 --sigma $s; done

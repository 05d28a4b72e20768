#!/bin/bash
one() { python bench.py --no-cpu-baseline --steps 1000 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['spmv_mode'],'sigma',d['config']['sigma'],'us',d['roofline']['launch_us'])"; }
for s in 5 16; do
echo "== sigma $s"; echo -n "full:        "; one --mode fused --sigma $s
for a in 3 7 15 31 63; do echo -n "ablate $a:    "; CSR5HIP_LIB=$PWD/scripts/probes/libcsr5hip_ablate$a.so one --mode fused --sigma $s; done
done

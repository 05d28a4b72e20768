#!/bin/bash
one() { python bench.py --no-cpu-baseline --steps 1000 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['spmv_mode'],'sigma',d['config']['sigma'],'GFLOPS',d['value'],'us',d['roofline']['launch_us'])"; }
echo "default env"; one --mode fused; one --mode fused; one --mode two-pass
echo "HIP_FORCE_DEV_KERNARG=1"; HIP_FORCE_DEV_KERNARG=1 one --mode fused
echo "HIP_FORCE_DEV_KERNARG=0"; HIP_FORCE_DEV_KERNARG=0 one --mode fused

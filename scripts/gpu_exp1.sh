#!/bin/bash
one() { python bench.py --no-cpu-baseline --steps 500 "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['spmv_mode'],'sigma',d['config']['sigma'],'GFLOPS',d['value'],'us',d['roofline']['launch_us'])"; }
echo "rowcap 353 (default)"; one --mode fused; one --mode fused --sigma 8; one --mode two-pass
export CSR5_BENCH_ROWCAP=60
echo "rowcap 60"; one --mode fused; one --mode fused --sigma 8; one --mode two-pass

#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -1
for w in scircuit webbase nd24k rmat22; do
  python bench.py --no-cpu-baseline --workload $w --steps 20 --warmup 3 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$w', 'convert ms', d['config']['csr_to_csr5_ms'], 'spmv us', d['roofline']['launch_us'], 'xwin', d['config']['lds_x_window'], d['config']['x_window_cover_pct'])"
done

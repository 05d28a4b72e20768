#!/bin/bash
# quick loop: parity tests, then bench lines for both modes on the headline workload
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -15
for mode in fused two-pass; do
  python bench.py --mode $mode --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$mode','sigma',d['config']['sigma'],'GFLOPS',d['value'],'us/step',round(d['ms_per_step']*1e3,3),'frac',d['roofline']['frac'])"
done
for s in 4 6 8 10 12 16 20; do
  python bench.py --mode fused --sigma $s --no-cpu-baseline --steps 500 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fused sigma',d['config']['sigma'],'GFLOPS',d['value'],'us',d['roofline']['launch_us'],'frac',d['roofline']['frac'])"
done
python bench.py --workload webbase --no-cpu-baseline --steps 300 2>&1 | tail -1 | cut -c1-400
python bench.py --workload nd24k --no-cpu-baseline --steps 200 2>&1 | tail -1 | cut -c1-400

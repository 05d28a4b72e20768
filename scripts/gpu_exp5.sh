#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); c=d['config']; print(c['workload'][:14], d['dtype'],'sigma',c['sigma'],'xwin',c['lds_x_window'],c['x_window_tiles'],'/',c['tiles'],'GFLOPS',d['value'],'us',d['roofline']['launch_us'],'GB/s',d['roofline']['achieved'])"; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -5
for xw in off force; do
echo "== x-window $xw"
for s in 12 16 20; do one --workload nd24k --steps 100 --sigma $s --x-window $xw; done
one --workload nd24k --dtype f64 --steps 100 --sigma 16 --x-window $xw
one --workload scircuit --steps 500 --x-window $xw
one --workload scircuit --steps 500 --sigma 8 --x-window $xw
one --workload webbase --steps 200 --x-window $xw
one --workload rmat22 --steps 30 --warmup 3 --x-window $xw
done
echo "== auto"; one --workload nd24k --steps 100; one --workload scircuit --steps 500; one --workload webbase --steps 200

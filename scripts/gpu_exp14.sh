#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); c=d['config']
print(c['workload'][:24], d['dtype'], 'sigma',c['sigma'],'xwin',int(c['lds_x_window']),'cover',c['x_window_cover_pct'],'% convert_ms',c['csr_to_csr5_ms'],' us',d['roofline']['launch_us'],'frac',d['roofline']['frac'])"; }
timeout 900 python -m pytest tests -m gpu -x -q -k "window or fuzz" 2>&1 | tail -3
one --steps 500
one --workload webbase --steps 200
one --workload nd24k --steps 100
one --workload nd24k --dtype f64 --steps 100
one --workload rmat22 --steps 30 --warmup 3
one --workload rmat24 --steps 10 --warmup 2

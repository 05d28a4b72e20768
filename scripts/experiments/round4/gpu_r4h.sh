#!/bin/bash
# round 4, call h: the whole gpu suite, then the default bench line
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
timeout 600 python bench.py > gpurun_out/r4h_bench.json 2> gpurun_out/r4h_bench.err; tail -c 600 gpurun_out/r4h_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4h_bench.json'))
r=d['roofline']; print('headline', d['value'], 'GFLOPS', d['ms_per_step'], 'ms', 'frac', r['frac'], 'live', r.get('x_live'))
for c in d.get('configs', []):
    rr=c.get('roofline', {}); print(c.get('workload','')[:30], c.get('value'), rr.get('frac'), rr.get('launch_us'), (rr.get('warm') or {}).get('frac'))
print(d.get('cpu_baseline'))
PY

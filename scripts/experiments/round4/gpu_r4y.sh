#!/bin/bash
# round 4, call y: smoke, the whole gpu suite, the round's profiles on the final sources, default bench
mkdir -p gpurun_out
timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | tail -3
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
ROUND=r04 timeout 2400 bash scripts/gpu_profiles_round.sh 2>&1 | grep -E "^rmat|^webbase|^scircuit|^nd24k" | cut -c1-200
timeout 600 python bench.py > gpurun_out/r4y_bench.json 2> gpurun_out/r4y_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4y_bench.json'))
r=d['roofline']; print('headline', d['value'], 'GFLOPS', d['ms_per_step'], 'ms', 'frac', r['frac'], 'traffic', r.get('traffic'), r.get('traffic_source'), 'live', (r.get('x_live') or {}).get('launch_us'))
for c in d.get('configs', []):
    rr=c.get('roofline', {}); print(c.get('workload','')[:30], c.get('value'), rr.get('frac'), rr.get('launch_us'), (rr.get('warm') or {}).get('frac'), rr.get('traffic'))
print(d.get('cpu_baseline'))
PY

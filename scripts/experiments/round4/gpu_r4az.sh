#!/bin/bash
# round 4, call az: final validation on the final sources: smoke, whole gpu suite, fuzz campaigns, the round's profiles, default bench
mkdir -p gpurun_out
timeout 600 python -c 'import __graft_entry__ as g; g.smoke()' 2>&1 | grep "smoke" | tail -3
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | grep -E "passed|failed|rror" | tail -3
CSR5_FUZZ_SEED=4242 CSR5_FUZZ_CASES=3000 timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -k seeded_fuzz 2>&1 | grep -E "passed|failed|rror" | tail -2
ROUND=r04 timeout 2400 bash scripts/gpu_profiles_round.sh 2>&1 | grep -E "^rmat|^webbase|^scircuit|^nd24k" | cut -c1-200
timeout 600 python bench.py > gpurun_out/r4y_bench.json 2> gpurun_out/r4y_bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/r4y_bench.json'))
r=d['roofline']; print('headline', d['value'], 'GFLOPS', d['ms_per_step'], 'ms', 'frac', r['frac'], 'traffic', r.get('traffic'), r.get('traffic_source'), 'live', (r.get('x_live') or {}).get('launch_us'), 'narrowed', (r.get('narrowed_values') or {}).get('launch_us'), (r.get('narrowed_values') or {}).get('y_bit_identical_to_headline_run'))
for c in d.get('configs', []):
    rr=c.get('roofline', {}); print(c.get('workload','')[:30], c.get('value'), rr.get('frac'), rr.get('launch_us'), (rr.get('warm') or {}).get('frac'), rr.get('traffic'))
print(d.get('cpu_baseline'))
PY

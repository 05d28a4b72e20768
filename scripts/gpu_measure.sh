set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for mode in fused two-pass; do
  python bench.py --mode $mode 2>&1 | tail -1 | tee gpurun_out/bench_scircuit_$mode.json
done
python bench.py --mode fused --launch eager --no-cpu-baseline 2>&1 | tail -1 | tee gpurun_out/bench_scircuit_eager.json
for s in 4 5 6 8 12 16 24 32; do
  python bench.py --mode fused --sigma $s --no-cpu-baseline --steps 500 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('sigma',d['config']['sigma'],'GFLOPS',d['value'],'us',d['roofline']['launch_us'],'frac',d['roofline']['frac'])"
done | tee gpurun_out/sigma_sweep_scircuit.txt
python bench.py --workload webbase --no-cpu-baseline --steps 300 2>&1 | tail -1 | tee gpurun_out/bench_webbase.json
python bench.py --workload rmat22 --no-cpu-baseline --steps 50 --warmup 5 2>&1 | tail -1 | tee gpurun_out/bench_rmat22.json

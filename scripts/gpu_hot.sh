#!/bin/bash
# LDS hot table of the slab kernel: parity tests, then bench lines hot on / off
one() { python bench.py --no-cpu-baseline --no-sub-configs "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
mkdir -p gpurun_out
python - <<'PY'
import torch
p = torch.cuda.get_device_properties(0)
print({k: getattr(p, k) for k in dir(p) if 'shared' in k.lower() or 'multi_processor' in k.lower()})
PY
timeout 1500 python -m pytest tests/test_gpu_slabs.py -x -q 2>&1 | tail -12
for w in rmat20 rmat22; do
  one --workload $w --steps 30 --warmup 3 --slabs 8 --slab-hot off
  for s in 8 16 32; do one --workload $w --steps 30 --warmup 3 --slabs $s --slab-hot force; done
done
one --workload rmat22 --steps 30 --warmup 3 --slabs 16 --slab-hot force --slab-shift 0
one --workload rmat24 --steps 10 --warmup 2 --slabs 16 --slab-hot off
for s in 16 32 64; do one --workload rmat24 --steps 10 --warmup 2 --slabs $s --slab-hot force; done
one --workload webbase --steps 200 --slabs 8 --slab-hot force

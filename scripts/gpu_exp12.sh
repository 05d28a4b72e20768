#!/bin/bash
one() { python bench.py --no-cpu-baseline "$@" 2>&1 | tail -1 | python scripts/benchline.py; }
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
echo "== tuned vs rule"
for w in scircuit webbase nd24k rmat22; do
  st=300; [ $w = rmat22 ] && st=30
  one --workload $w --steps $st --warmup 5; one --workload $w --steps $st --warmup 5 --sigma tuned
done
one --workload nd24k --dtype f64 --steps 100 --sigma tuned
python -c "
import sys; sys.path.insert(0,'.')
from benchmark_spmv_using_csr5_amd import matrices as M
import numpy as np
mat=M.example_matrix(); mat.val[:]=1; M.write_mtx('/tmp/example.mtx', mat)"
CSR5_SEED=3 CSR5_SIGMA=tuned benchmark_spmv_using_csr5_amd/csrc/spmv /tmp/example.mtx | tail -16

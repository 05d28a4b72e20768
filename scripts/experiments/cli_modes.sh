#!/bin/bash
# Is the per-process step-time mode a property of python/torch processes?  Same matrix through the C++ CLI, 10 processes.
python - <<'PY'
import sys; sys.path.insert(0, ".")
from benchmark_spmv_using_csr5_amd import matrices as M
mat = M.scircuit_like(); mat.val[:] = 1.0
M.write_mtx("/tmp/scircuit_like.mtx", mat)
PY
for i in 1 2 3 4 5 6 7 8 9 10; do
  CSR5_SEED=1 benchmark_spmv_using_csr5_amd/csrc/spmv /tmp/scircuit_like.mtx | grep "hipGraph replay" | cut -c1-70
done

#!/bin/bash
# webbase-like: slab count / shift sweep (cold and warm us per SpMV)
for s in auto 2 4 8 16; do
python - <<PY
import sys
sys.path.insert(0, "scripts/experiments/round5"); sys.path.insert(0, ".")
import numpy as np, torch
import walk_ab as W
from benchmark_spmv_using_csr5_amd import matrices as M
dev = torch.device("cuda", 0)
mat = M.webbase_like()
a = W.base_args(slabs="$s", tile_walk="off")
warm, cold, desc, b = W.measure(mat, "webbase", "f64", a, dev)
print("slabs $s warm %.2f (%.3f) cold %.2f (%.3f) %s" % (warm, b/(warm*1e-6)/8e12, cold, b/(cold*1e-6)/8e12, desc))
PY
done

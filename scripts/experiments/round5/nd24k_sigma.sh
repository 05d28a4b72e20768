for s in ${SIGMAS:-16 20 24 32 12}; do
python - <<PY
import sys, types
sys.path.insert(0, "scripts/experiments/round5"); sys.path.insert(0, ".")
import numpy as np, torch
import bench as B, walk_ab as W
from benchmark_spmv_using_csr5_amd import matrices as M
dev = torch.device("cuda", 0)
mat = M.nd24k_like(dtype=np.float32)
a = W.base_args(sigma="$s", slabs="0", tile_walk="off", x_window="force")
warm, cold, desc, b = W.measure(mat, "nd24k", "f32", a, dev)
print("sigma $s warm %.2f (%.3f) cold %.2f (%.3f) %s" % (warm, b/(warm*1e-6)/8e12, cold, b/(cold*1e-6)/8e12, desc))
PY
done

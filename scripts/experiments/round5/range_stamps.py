"""When does every wavefront range of k_spmv_range start and end?  (experiment build -DCSR5_RANGE_STAMPS through CSR5HIP_LIB)
R-MAT at the library's defaults: per slab the spread of the 256 ranges' durations, per workgroup the wait at the slab boundary,
per XCD the end of its walk."""
import ctypes as C
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import bench as B  # noqa: E402
from benchmark_spmv_using_csr5_amd import _capi  # noqa: E402
from benchmark_spmv_using_csr5_amd import matrices as M  # noqa: E402

scale = int(sys.argv[1]) if len(sys.argv) > 1 else 24
dev = torch.device("cuda", 0)
a = types.SimpleNamespace(values="int", sigma="-1", mode="fused", x_window="auto", xcd_remap=1, lds_y="auto", stream_nt="auto",
                          slabs="auto", slab_shift=None, slab_hot="auto", x_snapshot=0, zero_empty=0, tile_walk="off", walk_ranges=0, seed=1)
mat = M.rmat_device(scale, 16, seed=5, rank=0, world=1, device=dev)
prob = B.Problem(mat, f"rmat{scale}", "f64", a, dev, 14)
S = prob.info.column_slabs
for _ in range(3):
    prob.A.spmv(1.0, prob.yd)
torch.cuda.synchronize()
lib = _capi.load()
n = S * 256
buf = np.zeros(2 * n, dtype=np.uint64)
lib.csr5hip_debug_range_stamps.argtypes = [C.c_void_p, C.c_int]
assert lib.csr5hip_debug_range_stamps(buf.ctypes.data, 2 * n) == 0
t0 = buf[0::2].astype(np.int64).reshape(S, 256)
t1 = buf[1::2].astype(np.int64).reshape(S, 256)
base = t0[t0 > 0].min()
s, e = (t0 - base) / 100.0, (t1 - base) / 100.0
d = e - s
print(f"slabs {S}; kernel span {e.max():.1f} us")
for k in range(S):
    wg = d[k].reshape(32, 8)  # range rho = wg * 8 + wave
    print(f"slab {k:2d}: start {s[k].min():7.1f}..{s[k].max():7.1f} end {e[k].min():7.1f}..{e[k].max():7.1f}  range us: mean {d[k].mean():6.1f} min {d[k].min():6.1f} "
          f"max {d[k].max():6.1f} (max/mean {d[k].max() / d[k].mean():.2f}); per workgroup max-of-8: mean {wg.max(1).mean():6.1f} max {wg.max(1).max():6.1f}; "
          f"first / last quarter of the rows {d[k][:64].mean():6.1f} / {d[k][192:].mean():6.1f}")
# systematic or random?  mean duration by wavefront slot (0..7) and by workgroup (0..31), relative to the slab's mean
rel = d / d.mean(axis=1, keepdims=True)
byw = rel.reshape(S, 32, 8).mean(axis=(0, 1))
bywg = rel.reshape(S, 32, 8).mean(axis=(0, 2))
print("by wavefront slot:", np.round(byw, 3))
print("by workgroup     :", np.round(bywg, 3))
# the same (XCD, workgroup, wavefront) slot in the XCD's first and second slab: correlation of the relative durations
order = np.argsort(s.min(axis=1))  # slabs by start time: first-round slabs, then second-round ones
first, second = order[:S // 2], order[S // 2:]
print("first-round slabs", first, "second-round", second)

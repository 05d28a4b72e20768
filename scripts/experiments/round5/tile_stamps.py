"""When do the tiles of the one-tile kernel k_spmv start, get their data, and end?  (experiment build -DCSR5_TILE_STAMPS through
CSR5HIP_LIB; workloads with < 65 536 tiles)  COLD: five copies of the problem are multiplied in rotation, the stamps are those of
the last launch.  usage: tile_stamps.py nd24k|scircuit|webbase [warm]"""
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

w = sys.argv[1] if len(sys.argv) > 1 else "nd24k"
warm = len(sys.argv) > 2 and sys.argv[2] == "warm"
dev = torch.device("cuda", 0)
a = types.SimpleNamespace(values="int", sigma="-1", mode="fused", x_window="auto", xcd_remap=1, lds_y="auto", stream_nt="auto",
                          slabs="0", slab_shift=None, slab_hot="auto", x_snapshot=0, zero_empty=0, tile_walk="off", walk_ranges=0, seed=1)
dtype_name = "f32" if w == "nd24k" else "f64"
npd = np.float32 if dtype_name == "f32" else np.float64
mat = {"nd24k": lambda: M.nd24k_like(dtype=npd), "scircuit": lambda: M.scircuit_like(dtype=npd), "webbase": lambda: M.webbase_like(dtype=npd)}[w]()
k = 1 if warm else max(3, int(2 * 268435456 // (mat.nnz * (4 + (4 if dtype_name == "f32" else 8)))) + 2)
probs = [B.Problem(mat, w, dtype_name, a, dev, 14) for _ in range(min(k, 40))]
for rep in range(4):
    for p in probs:
        p.A.spmv(1.0, p.yd)
torch.cuda.synchronize()
lib = _capi.load()
tiles = probs[0].info.p - 1
buf = np.zeros(3 * tiles, dtype=np.uint64)
fetch = getattr(lib, "csr5hip_debug_tile_stamps_" + dtype_name)
fetch.argtypes = [C.c_void_p, C.c_int]
assert fetch(buf.ctypes.data, 3 * tiles) == 0
st = buf.astype(np.int64).reshape(tiles, 3)
ok = (st[:, 0] > 0) & (st[:, 2] >= st[:, 0])
base = st[ok, 0].min()
s, mid, e = (st[ok, 0] - base) / 100.0, (st[ok, 1] - base) / 100.0, (st[ok, 2] - base) / 100.0
has_mid = st[ok, 1] >= st[ok, 0]
print(f"{w} {'warm' if warm else 'cold (%d copies)' % len(probs)}: tiles {ok.sum()} of {tiles}; span {e.max():.1f} us; lifetime mean {np.mean(e - s):.2f} median {np.median(e - s):.2f} "
      f"p99 {np.percentile(e - s, 99):.2f}; start -> data landed: median {np.median((mid - s)[has_mid]):.2f}; data -> end: median {np.median((e - mid)[has_mid]):.2f}")
edges = np.linspace(0, e.max(), 17)
for lo, hi in zip(edges[:-1], edges[1:]):
    m = 0.5 * (lo + hi)
    print(f"  t = {m:6.1f} us: resident {int(((s <= m) & (e > m)).sum()):6d} ({((s <= m) & (e > m)).sum() / 256:.1f} per CU)  started in bin {int(((s >= lo) & (s < hi)).sum()):6d}")

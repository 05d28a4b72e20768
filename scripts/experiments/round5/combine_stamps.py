"""When do the wavefronts of k_slab_combine start and end?  (experiment build with -DCSR5_COMBINE_STAMPS through CSR5HIP_LIB)
R-MAT 24 at the library's defaults; prints the distribution of wavefront lifetimes, how many are resident over time, and the
slowest blocks."""
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
for _ in range(3):
    prob.A.spmv(1.0, prob.yd)
torch.cuda.synchronize()
lib = _capi.load()
nblk = (prob.m + 255) // 256
buf = np.zeros(2 * nblk, dtype=np.uint64)
lib.csr5hip_debug_combine_stamps.argtypes = [C.c_void_p, C.c_int]
rc = lib.csr5hip_debug_combine_stamps(buf.ctypes.data, 2 * nblk)
assert rc == 0, rc
t0, t1 = buf[0::2].astype(np.int64), buf[1::2].astype(np.int64)
ok = (t0 > 0) & (t1 >= t0)
base = t0[ok].min()
s, e = (t0[ok] - base) / 100.0, (t1[ok] - base) / 100.0  # us (100 MHz)
life = e - s
print(f"blocks {ok.sum()} of {nblk}; kernel span {e.max():.1f} us; lifetime us: mean {life.mean():.2f} median {np.median(life):.2f} "
      f"p90 {np.percentile(life, 90):.2f} p99 {np.percentile(life, 99):.2f} max {life.max():.2f}")
edges = np.linspace(0, e.max(), 21)
for lo, hi in zip(edges[:-1], edges[1:]):
    mid = 0.5 * (lo + hi)
    resident = int(((s <= mid) & (e > mid)).sum())
    started = int(((s >= lo) & (s < hi)).sum())
    print(f"  t = {mid:7.1f} us: resident wavefronts {resident:6d} ({resident / 256:.1f} per CU)   started in bin {started:6d}")
order = np.argsort(-life)[:8]
idx = np.flatnonzero(ok)
print("slowest blocks (block, start us, lifetime us):", [(int(idx[i]), round(float(s[i]), 1), round(float(life[i]), 1)) for i in order])
# lifetime by block index (row position)
for q in range(8):
    sel = slice(q * len(life) // 8, (q + 1) * len(life) // 8)
    print(f"  blocks {q}/8 of the rows: mean lifetime {life[sel].mean():.2f} us, mean start {s[sel].mean():.1f} us")
# phases of a block (experiment build): bounds arrived / partials arrived / adds done, relative to the wavefront's start
ph = np.zeros(4 * nblk, dtype=np.uint64)
if hasattr(lib, "csr5hip_debug_combine_phase"):
    lib.csr5hip_debug_combine_phase.argtypes = [C.c_void_p, C.c_int]
    if lib.csr5hip_debug_combine_phase(ph.ctypes.data, 4 * nblk) == 0:
        ph = ph.astype(np.int64).reshape(nblk, 4)
        good = ok & (ph[:, 0] >= t0) & (ph[:, 2] >= ph[:, 0])
        a = (ph[good, 0] - t0[good]) / 100.0
        b = (ph[good, 1] - ph[good, 0]) / 100.0
        c = (ph[good, 2] - ph[good, 1]) / 100.0
        dd = (t1[good] - ph[good, 2]) / 100.0
        print(f"phases (us, mean / median): bounds {a.mean():.2f} / {np.median(a):.2f}; partials (last round) {b.mean():.2f} / {np.median(b):.2f}; "
              f"adds {c.mean():.2f} / {np.median(c):.2f}; stores {dd.mean():.2f} / {np.median(dd):.2f}")

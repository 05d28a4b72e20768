"""Per-tile timeline of the persistent hot-table kernel (experiment; needs `make -C benchmark_spmv_using_csr5_amd/csrc timing`).
Stages (100 MHz wall clock, 10 ns ticks), one stamp per tile of the slab child by lane 0 of the wavefront that ran it:
0 tile start | 1 all stream loads issued | 2 column words back, gathers issued | 3 gathers back | 4 descriptor decode +
spill reduce | 5 flag walk + LDS puts | 6 cross-lane | 7 flush + carries done.

    python scripts/probes/hot_timeline.py [scale=22] [slabs=auto]
"""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from benchmark_spmv_using_csr5_amd import _capi, matrices as M
_capi.LIB_PATH = os.path.join(ROOT, "benchmark_spmv_using_csr5_amd", "libcsr5hip_timing.so")
from benchmark_spmv_using_csr5_amd import handle as H
scale = int(sys.argv[1]) if len(sys.argv) > 1 else 22
slabs = sys.argv[2] if len(sys.argv) > 2 else "auto"
dev = torch.device("cuda:0")
mat = M.rmat_device_shard(scale, 16, 1, 0, 1, dev)
g = torch.Generator(device=dev).manual_seed(7)
va = torch.randint(0, 10, (mat.nnz,), generator=g, device=dev).to(torch.float64)
xd = torch.randint(0, 10, (mat.n,), generator=g, device=dev).to(torch.float64)
yd = torch.zeros(mat.m, dtype=torch.float64, device=dev)
A = H.anonymouslibHandle(mat.m, mat.n)
A.inputCSR(mat.nnz, mat.row_ptr, mat.col.clone(), va); A.setX(xd); A.setSigma(-1)
A.setColumnSlabs(1 if slabs == "auto" else int(slabs)); A.asCSR5()
i = A.info()
assert i.slab_hot == 1, "no hot table on this matrix"
p = i.slab_tiles
buf = torch.zeros(p * 8, dtype=torch.int64, device=dev)
lib = _capi.load()
for _ in range(5):
    A.spmv(1.0, yd)
torch.cuda.synchronize()
A.timer_start(); A.spmv_repeat(1.0, yd, 10); us_plain = A.timer_stop() * 100
assert lib.csr5hip_debug_set_timing_buffer(C.c_void_p(buf.data_ptr())) == 0
A.timer_start(); A.spmv_repeat(1.0, yd, 3); us_probe = A.timer_stop() * 1e3 / 3
torch.cuda.synchronize()
ts = buf.cpu().numpy().reshape(p, 8).astype(np.int64)
ok = (ts[:, 7] > 0) & (ts[:, 0] > 0)
ts = ts[ok]
t0 = ts[:, 0].min()
print(f"R-MAT {scale}: child sigma={i.slab_sigma} tiles={p} stamped={ok.sum()} slabs={i.column_slabs} cover={i.slab_hot_cover_pct}%  "
      f"step {us_plain:.1f} us without stamps, {us_probe:.1f} us with; kernel span {(ts[:,7].max()-t0)/100:.1f} us")
names = ["issue stream loads", "col words back -> gathers issued", "gathers back", "decode + spill reduce",
         "flag walk + LDS puts", "cross-lane", "flush + carries"]
tot = 0
for k in range(7):
    d = (ts[:, k + 1] - ts[:, k]) * 10
    tot += d.mean()
    print("stage %d %-34s ns: mean %6.0f  p10 %5d p50 %5d p90 %5d max %6d" % ((k, names[k], d.mean()) + tuple(np.percentile(d, [10, 50, 90, 100]))))
life = (ts[:, 7] - ts[:, 0]) * 10
print("tile time ns: mean %.0f p10 %d p50 %d p90 %d max %d" % ((life.mean(),) + tuple(np.percentile(life, [10, 50, 90, 100]))))
print("tiles per wave-slot = %.1f ; sum of tile times / (256 CUs * 16 waves) = %.1f us" % (len(ts) / 4096, life.sum() / 4096 / 1e3))
# completion curve: when had x % of the tiles finished (share of the kernel span)?  A long tail = load imbalance between
# wavefronts / workgroups / XCDs under the static tile assignment.
end = np.sort(ts[:, 7] - t0).astype(np.float64)
span = end[-1]
print("completion curve (share of the span at which N % of the tiles were done):",
      {q: round(float(end[int(len(end) * q / 100) - 1] / span), 3) for q in (25, 50, 75, 90, 95, 98, 99, 100)})
start = np.sort(ts[:, 0] - t0).astype(np.float64)
print("start curve:", {q: round(float(start[int(len(start) * q / 100) - 1] / span), 3) for q in (25, 50, 75, 90, 95, 99, 100)})

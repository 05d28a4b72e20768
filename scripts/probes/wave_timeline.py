"""Per-wave timeline of the fused SpMV kernel (experiment; needs `make -C benchmark_spmv_using_csr5_amd/csrc timing`).
Stages (100 MHz wall clock, 10 ns ticks): 0 wave start | 1 tile loads returned enough to issue gathers |
2 gathers issued | 3 gathers returned | 4 decode done | 5 flag walk done | 6 cross-lane done | 7 end."""
import ctypes as C, os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from benchmark_spmv_using_csr5_amd import _capi, matrices as M
_capi.LIB_PATH = os.path.join(ROOT, "benchmark_spmv_using_csr5_amd", "libcsr5hip_timing.so")
from benchmark_spmv_using_csr5_amd import handle as H
sigma = int(sys.argv[1]) if len(sys.argv) > 1 else -1
mode = int(sys.argv[2]) if len(sys.argv) > 2 else 1
mat = M.scircuit_like()
val, x = M.fill_values(mat.nnz, mat.n, np.float64, 14, "int")
dev = "cuda:0"
rp, ci, va, xd = (torch.from_numpy(a).to(dev) for a in (mat.row_ptr, mat.col, val, x))
yd = torch.zeros(mat.m, dtype=torch.float64, device=dev)
A = H.anonymouslibHandle(mat.m, mat.n)
A.inputCSR(mat.nnz, rp, ci, va); A.setX(xd); A.setSigma(sigma); A.setSpmvMode(mode); A.asCSR5()
p = A.info().p
buf = torch.zeros(p * 8, dtype=torch.int64, device=dev)
lib = _capi.load()
for _ in range(20):
    A.spmv(1.0, yd)
torch.cuda.synchronize()
assert lib.csr5hip_debug_set_timing_buffer(C.c_void_p(buf.data_ptr())) == 0
A.spmv_repeat(1.0, yd, 5)   # back-to-back launches; the last one's stamps survive
torch.cuda.synchronize()
ts = buf.cpu().numpy().reshape(p, 8)[: p - 1].astype(np.int64)
ok = ts[:, 7] > 0
ts = ts[ok]
t0 = ts[:, 0].min()
print(f"sigma={A.info().sigma} tiles={p-1} stamped={ok.sum()} kernel span (first wave start -> last wave end) = {(ts[:,7].max()-t0)*10} ns")
print("wave start offset  ns: min %d  p50 %d  p90 %d  max %d" % tuple(np.percentile((ts[:,0]-t0)*10, [0,50,90,100])))
names = ["loads->gather issue", "gather issue", "gathers return", "decode+spill reduce", "flag walk+stores", "cross-lane", "final stores/carries"]
for k in range(7):
    d = (ts[:, k+1] - ts[:, k]) * 10
    print("stage %d %-22s ns: p10 %5d p50 %5d p90 %5d max %6d" % ((k, names[k]) + tuple(np.percentile(d, [10,50,90,100]))))
life = (ts[:,7]-ts[:,0])*10
print("wave lifetime ns: p10 %d p50 %d p90 %d max %d" % tuple(np.percentile(life,[10,50,90,100])))

"""Is the 5.7 / 6.3 us bimodality of the scircuit bench a per-process or a per-allocation property?
Several handles (fresh device allocations each) timed inside ONE process, with dummy allocations in between."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from benchmark_spmv_using_csr5_amd import matrices as M, handle as H
dev = "cuda:0"
mat = M.scircuit_like()
val, x = M.fill_values(mat.nnz, mat.n, np.float64, 14, "int")
junk = []
for rep in range(8):
    if rep % 2:
        junk.append(torch.empty((1 << 20) * (rep + 1) + 12345, dtype=torch.uint8, device=dev))
    rp, ci, va, xd = (torch.from_numpy(a).to(dev) for a in (mat.row_ptr, mat.col, val, x))
    yd = torch.zeros(mat.m, dtype=torch.float64, device=dev)
    A = H.anonymouslibHandle(mat.m, mat.n)
    A.inputCSR(mat.nnz, rp, ci, va); A.setX(xd); A.setSigma(-1); A.asCSR5()
    A.spmv_repeat(1.0, yd, 500)
    torch.cuda.synchronize()
    ts = []
    for k in range(5):
        A.timer_start(); A.spmv_repeat(1.0, yd, 500); ts.append(A.timer_stop() / 500 * 1e3)
    print(f"instance {rep}: us/step " + " ".join(f"{t:.3f}" for t in ts) + f"  val@{va.data_ptr():#x} y@{yd.data_ptr():#x}", flush=True)
    A.destroy(); A.close()
    del rp, ci, va, xd, yd

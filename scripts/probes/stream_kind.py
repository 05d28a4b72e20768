"""Launch-overhead probe: the same 500-launch hipGraph replay on the null stream, a non-blocking stream and a
high-priority stream (argv[1] = null | nonblocking | high).  Prints us per SpMV."""
import os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from benchmark_spmv_using_csr5_amd import matrices as M, handle as H
kind = sys.argv[1] if len(sys.argv) > 1 else "null"
dev = "cuda:0"
mat = M.scircuit_like()
val, x = M.fill_values(mat.nnz, mat.n, np.float64, 14, "int")
rp, ci, va, xd = (torch.from_numpy(a).to(dev) for a in (mat.row_ptr, mat.col, val, x))
yd = torch.zeros(mat.m, dtype=torch.float64, device=dev)
stream = None
if kind == "nonblocking":
    stream = torch.cuda.Stream(device=dev)
elif kind == "high":
    stream = torch.cuda.Stream(device=dev, priority=-1)
torch.cuda.synchronize()
A = H.anonymouslibHandle(mat.m, mat.n, stream=stream.cuda_stream if stream is not None else None)
A.inputCSR(mat.nnz, rp, ci, va); A.setX(xd); A.setSigma(-1); A.asCSR5()
for _ in range(40):
    A.spmv_repeat(1.0, yd, 500)
torch.cuda.synchronize()
ts = []
for k in range(5):
    A.timer_start(); A.spmv_repeat(1.0, yd, 500); ts.append(A.timer_stop() / 500 * 1e3)
print(f"{kind:12s} us/step " + " ".join(f"{t:.3f}" for t in ts), flush=True)
A.destroy(); A.close()

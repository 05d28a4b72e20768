"""Does the step time depend on WHO allocated the arrays?  Same process, same handle flow:
A = torch tensors (caching allocator), B = csr5hip_malloc (plain hipMalloc per array).  Prints us per SpMV."""
import ctypes as C, os, sys, numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from benchmark_spmv_using_csr5_amd import _capi, matrices as M, handle as H
dev = "cuda:0"
lib = _capi.load()
mat = M.scircuit_like()
val, x = M.fill_values(mat.nnz, mat.n, np.float64, 14, "int")


def run(ptrs):
    rp, ci, va, xd, yd = ptrs
    A = H.anonymouslibHandle(mat.m, mat.n)
    A.inputCSR(mat.nnz, rp, ci, va); A.setX(xd); A.setSigma(-1); A.asCSR5()
    for _ in range(4):
        A.spmv_repeat(1.0, yd, 500)
    lib.csr5hip_synchronize()
    ts = []
    for k in range(4):
        A.timer_start(); A.spmv_repeat(1.0, yd, 500); ts.append(A.timer_stop() / 500 * 1e3)
    A.destroy(); A.close()
    return min(ts)


def raw(arr):
    p = C.c_void_p()
    assert lib.csr5hip_malloc(C.byref(p), arr.nbytes) == 0
    assert lib.csr5hip_memcpy_h2d(p, arr.ctypes.data, arr.nbytes) == 0
    return p.value


for rep in range(3):
    t = [torch.from_numpy(a).to(dev) for a in (mat.row_ptr, mat.col, val, x)]
    y = torch.zeros(mat.m, dtype=torch.float64, device=dev)
    ta = run(t + [y])
    r = [raw(a) for a in (mat.row_ptr, mat.col, val, x, np.zeros(mat.m))]
    tb = run(r)
    for p in r:
        lib.csr5hip_device_free(C.c_void_p(p))
    print(f"torch tensors {ta:.3f} us   csr5hip_malloc {tb:.3f} us", flush=True)

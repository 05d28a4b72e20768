"""How do very long rows behave?  One 10 M-nnz row + 200 k short rows; us per SpMV in both modes."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from benchmark_spmv_using_csr5_amd import handle as H, matrices as M
rng = np.random.default_rng(1)
for big in (1_000_000, 10_000_000):
    lens = rng.integers(1, 9, size=200_000)
    lens[1000] = big
    mat = M.csr_from_row_lengths(lens, 1_000_000, rng, name="dense-row")
    val, x = M.fill_values(mat.nnz, mat.n, np.float64, 3, "int")
    dev = "cuda:0"
    rp, ci, va, xd = (torch.from_numpy(a).to(dev) for a in (mat.row_ptr, mat.col, val, x))
    yd = torch.zeros(mat.m, dtype=torch.float64, device=dev)
    ref = torch.sparse_csr_tensor(rp.long(), ci.long(), va, size=(mat.m, mat.n)) @ xd
    for mode in (H.SPMV_FUSED, H.SPMV_TWO_PASS):
        for sigma in (4, 16):
            A = H.anonymouslibHandle(mat.m, mat.n)
            A.inputCSR(mat.nnz, rp, ci, va); A.setX(xd); A.setSigma(sigma); A.setSpmvMode(mode)
            assert A.asCSR5() == 0
            for _ in range(3): A.spmv(1.0, yd)
            torch.cuda.synchronize()
            ok = bool(torch.equal(yd, ref))
            A.timer_start()
            for _ in range(20): A.spmv(1.0, yd)
            ms = A.timer_stop()
            print(f"row={big} nnz={mat.nnz} mode={'fused' if mode else 'two-pass'} sigma={sigma} tiles={A.info().p}: {ms/20*1e3:9.1f} us/SpMV exact={ok}")
            A.destroy(); A.close()

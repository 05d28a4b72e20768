"""Size spread of the column slabs of R-MAT 20 / 22 / 24 under the xor-fold hash (max and min slab over the mean; for
16 slabs also two consecutive slabs per XCD): why the hot kernel deals slabs to XCDs by size."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from benchmark_spmv_using_csr5_amd import matrices as M
dev = torch.device("cuda:0")
for scale, S in ((22, 8), (24, 16), (20, 8)):
    mat = M.rmat_device_shard(scale, 16, 1, 0, 1, dev)
    bits = S.bit_length() - 1
    v = (mat.col.to(torch.int64) >> 4)
    out = torch.zeros_like(v)
    while int(v.max()) > 0:
        out ^= v & (S - 1)
        v >>= bits
    cnt = torch.bincount(out, minlength=S).cpu().numpy()
    print(scale, S, "slab nnz max/mean = %.3f" % (cnt.max() / cnt.mean()), "min/mean = %.3f" % (cnt.min() / cnt.mean()))
    if S == 16:
        pair = cnt.reshape(8, 2).sum(1)
        print("   per XCD (2 consecutive slabs): max/mean = %.3f" % (pair.max() / pair.mean()))
        # what the library does (csr5_capi.hip build_slabs): longest slab first onto the XCD with the least work that has a round free
        load, used = [0] * 8, [0] * 8
        for k in sorted(range(S), key=lambda k: -cnt[k]):
            x = min((x for x in range(8) if used[x] < 2), key=lambda x: load[x])
            load[x] += int(cnt[k]); used[x] += 1
        print("   per XCD (dealt longest first): max/mean = %.4f  min/mean = %.4f" % (max(load) / (sum(load) / 8), min(load) / (sum(load) / 8)))

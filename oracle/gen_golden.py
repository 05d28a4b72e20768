#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE itself (TEST INFRASTRUCTURE ONLY).

Runs only where oracle/_ref/*.so exist, i.e. in the build container where /root/reference is mounted
(`make -C oracle ref`).  Every array written here was produced by the reference's own code:

  * fmt_w{omega}_s{sigma}_*  -- tile_ptr, tile_desc, offset_ptr, offset, tile-transposed column_index
    and value from CSR5_avx2/detail/avx2/format_avx2.h re-instantiated with ANONYMOUSLIB_CSR5_OMEGA =
    omega (oracle/ref_format.cpp);
  * y_avx2_{int,real}        -- y from the real CSR5_avx2 handle (omega 4, sigma 16, fp64;
    oracle/ref_spmv.cpp) on the reference CLI's integer data and on uniform(-1,1) data.

Inputs (row_ptr, col, val, x) are stored next to the outputs, so the fixtures are self-contained data;
no reference source text is stored anywhere.  Re-run:  python oracle/gen_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from benchmark_spmv_using_csr5_amd import matrices as M  # noqa: E402
from oracle.csr5_oracle import Reference  # noqa: E402
from tests import zoo  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FORMATS = [(64, 4), (64, 16), (64, 24), (32, 8), (4, 16)]
MAX_NNZ = 13000  # keep the committed fixtures small


def main():
    assert Reference.available() and all(Reference.available(w) for w, _ in FORMATS), \
        "oracle/_ref is missing: run `make -C oracle ref` where /root/reference exists"
    ref = Reference()
    os.makedirs(OUT, exist_ok=True)
    index = []
    for mat in zoo.small_zoo():
        if mat.nnz > MAX_NNZ:
            continue
        val_i, x_i = M.fill_values(mat.nnz, mat.n, np.float64, seed=5, mode="int")
        val_r, x_r = M.fill_values(mat.nnz, mat.n, np.float64, seed=9, mode="real")
        d = dict(m=mat.m, n=mat.n, row_ptr=mat.row_ptr, col=mat.col, val_int=val_i, x_int=x_i,
                 val_real=val_r, x_real=x_r)
        for omega, sigma in FORMATS:
            f = ref.convert(omega, sigma, mat.m, mat.row_ptr, mat.col, val_i)
            k = f"fmt_w{omega}_s{sigma}_"
            d[k + "params"] = np.array([f.bit_y, f.bit_ss, f.num_packet, f.p, f.num_offsets, f.tail_start])
            d[k + "tile_ptr"] = f.tile_ptr
            d[k + "tile_desc"] = f.tile_desc
            d[k + "offset_ptr"] = f.offset_ptr
            d[k + "offset"] = f.offset
            d[k + "col"] = f.col
            d[k + "val"] = f.val
        y0 = np.full(mat.m, 777.0)  # poison: rows the reference does not write keep it
        d["y_avx2_int"], _, _ = ref.avx2_spmv(mat.m, mat.n, mat.row_ptr, mat.col, val_i, x_i, y0=y0)
        d["y_avx2_real"], _, _ = ref.avx2_spmv(mat.m, mat.n, mat.row_ptr, mat.col, val_r, x_r, y0=y0)
        name = mat.name.split("(")[0].replace("-", "_")
        np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
        index.append(name)
        print(f"{name:24s} m={mat.m:6d} nnz={mat.nnz:6d}")
    with open(os.path.join(OUT, "INDEX.txt"), "w") as f:
        f.write("\n".join(index) + "\n")


if __name__ == "__main__":
    main()

#!/usr/bin/env python3
"""Writes the Matrix Market fixtures under tests/golden/mtx/ and the CSR the REFERENCE's own ingest
(CSR5_avx2/main.cpp:126-281 through oracle/_ref/libref_ingest.so) builds from each of them.

Run in the build container only (needs /root/reference):  python oracle/gen_golden_mtx.py
The .mtx files are our own seeded data; expected.npz holds, per fixture, m, n, row_ptr, col, val
exactly as the reference CLI held them at main.cpp:281, or the CLI's exit code for the broken files.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.csr5_oracle import MtxExit, Reference, build_oracle  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden", "mtx")


def entries(rng, m, n, nz, lower=False, diag_share=0.2):
    r = rng.integers(1, m + 1, nz)
    c = rng.integers(1, n + 1, nz)
    if lower:
        r, c = np.maximum(r, c), np.minimum(r, c)
        d = rng.random(nz) < diag_share
        c = np.where(d, r, c)
    return r, c


def fmt_real(rng, k):
    v = rng.standard_normal(k) * 10.0 ** rng.integers(-8, 9, k)
    styles = rng.integers(0, 5, k)
    out = []
    for x, s in zip(v, styles):
        if s == 0:
            out.append(repr(float(x)))
        elif s == 1:
            out.append(f"{x:.6e}")
        elif s == 2:
            out.append(f"{x:.17g}")
        elif s == 3:
            out.append(f"{x:+.3f}")
        else:
            out.append(f"{x:.12E}")
    return out


def write(name, text, newline="\n"):
    with open(os.path.join(OUT, name), "w", newline="") as f:
        f.write(text.replace("\n", newline))


def main():
    os.makedirs(OUT, exist_ok=True)
    build_oracle()
    rng = np.random.default_rng(20240928)
    files = {}

    r, c = entries(rng, 40, 40, 300)
    v = fmt_real(rng, 300)
    files["general_real.mtx"] = "%%MatrixMarket matrix coordinate real general\n% a comment\n%another\n40 40 300\n" + \
        "".join(f"{a} {b} {x}\n" for a, b, x in zip(r, c, v))

    r, c = entries(rng, 64, 64, 400, lower=True)
    v = fmt_real(rng, 400)
    files["symmetric_real.mtx"] = "%%MatrixMarket matrix coordinate real symmetric\n64 64 400\n" + \
        "".join(f"{a}\t{b}  {x}\n" for a, b, x in zip(r, c, v))

    r, c = entries(rng, 30, 30, 120, lower=True)
    files["hermitian_real.mtx"] = "%%MatrixMarket matrix coordinate real Hermitian\n30 30 120\n" + \
        "".join(f"{a} {b} {a * 0.5 - b}\n" for a, b in zip(r, c))

    r, c = entries(rng, 30, 30, 120, lower=True, diag_share=0.0)
    files["skew_real.mtx"] = "%%MatrixMarket matrix coordinate real skew-symmetric\n30 30 120\n" + \
        "".join(f"{a} {b} {a - b}.25\n" for a, b in zip(r, c))

    r, c = entries(rng, 50, 70, 333)
    files["pattern_general.mtx"] = "%%MatrixMarket matrix coordinate pattern general\n%\n50 70 333\n" + \
        "".join(f"{a} {b}\n" for a, b in zip(r, c))

    r, c = entries(rng, 45, 45, 200, lower=True)
    files["pattern_symmetric.mtx"] = "%%MatrixMarket MATRIX Coordinate PATTERN Symmetric\n45 45 200\n" + \
        "".join(f"{a} {b}\n" for a, b in zip(r, c))

    r, c = entries(rng, 33, 21, 150)
    iv = rng.integers(-50, 51, 150)
    files["integer_general.mtx"] = "%%MatrixMarket matrix coordinate integer general\n33 21 150\n" + \
        "".join(f"{a} {b} {x}\n" for a, b, x in zip(r, c, iv))

    r, c = entries(rng, 25, 25, 90, lower=True)
    iv = rng.integers(0, 10, 90)
    files["integer_symmetric.mtx"] = "%%MatrixMarket matrix coordinate integer symmetric\n25 25 90\n" + \
        "".join(f"{a} {b} {x}\n" for a, b, x in zip(r, c, iv))

    # rows 10..29 empty, one dense row, duplicates, blank lines between entries, trailing blanks
    rows = np.concatenate([rng.integers(1, 10, 60), np.full(80, 35), rng.integers(30, 61, 60), [3, 3, 3]])
    cols = np.concatenate([rng.integers(1, 61, 60), rng.integers(1, 61, 80), rng.integers(1, 61, 60), [7, 7, 7]])
    perm = rng.permutation(rows.size)
    rows, cols = rows[perm], cols[perm]
    body = ""
    for k, (a, b) in enumerate(zip(rows, cols)):
        body += f"  {a}   {b}   {k * 0.125}  \n"
        if k % 17 == 0:
            body += "\n"
    files["empty_rows_blank_lines.mtx"] = f"%%MatrixMarket matrix coordinate real general\n60 60 {rows.size}\n" + body

    r, c = entries(rng, 20, 20, 64)
    files["crlf.mtx"] = ("%%MatrixMarket matrix coordinate real general\n% dos line ends\n20 20 64\n" +
                         "".join(f"{a} {b} {a + b / 16}\n" for a, b in zip(r, c)), "\r\n")

    # size line after a blank line; no newline at the end of the file
    r, c = entries(rng, 12, 12, 30)
    files["blank_before_size.mtx"] = ("%%MatrixMarket matrix coordinate real general\n%c\n\n12 12 30\n" +
                                      "".join(f"{a} {b} {a * b}\n" for a, b in zip(r, c))).rstrip("\n")

    # entries spread freely over lines: only a token scanner reads this (fscanf semantics)
    r, c = entries(rng, 16, 16, 40)
    toks = []
    for a, b in zip(r, c):
        toks += [str(a), str(b), f"{a / (b + 1):.5f}"]
    body = ""
    for k, t in enumerate(toks):
        body += t + ("\n" if k % 7 == 6 else " ")
    files["free_form_tokens.mtx"] = "%%MatrixMarket matrix coordinate real general\n16 16 40\n" + body + "\n"

    # more lines than announced entries: only the first nz count
    r, c = entries(rng, 10, 10, 25)
    files["extra_lines.mtx"] = "%%MatrixMarket matrix coordinate integer general\n10 10 20\n" + \
        "".join(f"{a} {b} {a}\n" for a, b in zip(r, c))

    files["empty_matrix.mtx"] = "%%MatrixMarket matrix coordinate real general\n5 5 0\n"
    files["one_entry.mtx"] = "%%MatrixMarket matrix coordinate real symmetric\n3 3 1\n3 1 2.5\n"
    files["special_values.mtx"] = "%%MatrixMarket matrix coordinate real general\n4 4 8\n" \
        "1 1 1e308\n1 2 -1E-320\n2 1 0.1\n2 2 123456789012345678901234567890\n3 3 -0\n3 4 4.9406564584124654e-324\n" \
        "4 1 .5\n4 4 5.\n"

    # broken files: the CLI's exit codes
    files["err_banner.mtx"] = "%MatrixMarket matrix coordinate real general\n2 2 1\n1 1 1\n"
    files["err_banner_short.mtx"] = "%%MatrixMarket matrix coordinate real\n2 2 1\n1 1 1\n"
    files["err_field.mtx"] = "%%MatrixMarket matrix coordinate quaternion general\n2 2 1\n1 1 1\n"
    files["err_complex.mtx"] = "%%MatrixMarket matrix coordinate complex general\n2 2 1\n1 1 1 0\n"
    files["err_size.mtx"] = "%%MatrixMarket matrix coordinate real general\n% only comments\n"

    ref = Reference()
    expected = {}
    for name, text in files.items():
        if isinstance(text, tuple):
            write(name, text[0], text[1])
        else:
            write(name, text)
        key = name[:-4]
        try:
            m, n, row_ptr, col, val = ref.ingest(os.path.join(OUT, name))
            expected[key + ".code"] = np.int32(0)
            expected[key + ".dims"] = np.array([m, n], dtype=np.int32)
            expected[key + ".row_ptr"] = row_ptr
            expected[key + ".col"] = col
            expected[key + ".val"] = val
            print(f"{name:32s} m={m} n={n} nnz={col.size}")
        except MtxExit as e:
            expected[key + ".code"] = np.int32(e.code)
            print(f"{name:32s} exit {e.code}")
    np.savez_compressed(os.path.join(OUT, "expected.npz"), **expected)


if __name__ == "__main__":
    main()

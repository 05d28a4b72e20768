#!/usr/bin/env python3
"""Resource table of the SpMV kernels from hipcc's own -Rpass-analysis=kernel-resource-usage remarks.

    python scripts/kernel_resources.py [--filter k_spmv] [--sigmas few|all] > profiles/rNN_resources.md
Compiles csr5_spmv.hip, csr5_slab.hip and csr5_hot.hip for gfx950 with the remark pass on and prints, per kernel instantiation,
VGPRs / SGPRs / scratch / LDS / occupancy in waves per SIMD.  Runs without a GPU."""
import argparse
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "benchmark_spmv_using_csr5_amd", "csrc")


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--filter", default="k_spmv|k_slab|k_calibrate|k_range|k_x_permute")
    ap.add_argument("--sigmas", default="few")
    ap.add_argument("--match", default=None, help="regex on the demangled name")
    args = ap.parse_args()
    rows = []
    for src, defs in (("csr5_spmv.hip", ["-DCSR5_SPMV_ONLY_F64"]), ("csr5_spmv.hip", ["-DCSR5_SPMV_ONLY_F32"]),
                      ("csr5_slab.hip", []), ("csr5_hot.hip", [])):
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics",
               f"-I{ROOT}/include", "-Rpass-analysis=kernel-resource-usage", "-c", os.path.join(CSRC, src),
               "-o", "/dev/null"] + defs + (["-DCSR5_FEW_SIGMAS"] if args.sigmas == "few" else [])
        txt = subprocess.run(cmd, capture_output=True, text=True).stderr
        for blk in re.split(r"remark: [^\n]*Function Name: ", txt)[1:]:
            name = blk.split("\n")[0].strip()

            def f(key):
                m = re.search(key + r": (\d+)", blk)
                return int(m.group(1)) if m else -1
            rows.append([name, f("VGPRs"), f("AGPRs"), f("SGPRs"), f(r"ScratchSize \[bytes/lane\]"),
                         f(r"LDS Size \[bytes/block\]"), f(r"Occupancy \[waves/SIMD\]")])
    names = demangle([r[0] for r in rows])
    print("| kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | static LDS B | waves/SIMD (register-limited) |")
    print("|---|---|---|---|---|---|---|")
    seen = set()
    for r, n in zip(rows, names):
        n = re.sub(r"^void ", "", n).split("(")[0]
        if not re.search(args.filter, n) or (args.match and not re.search(args.match, n)) or n in seen:
            continue
        seen.add(n)
        print(f"| `{n}` | {r[1]} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]} |")


if __name__ == "__main__":
    main()

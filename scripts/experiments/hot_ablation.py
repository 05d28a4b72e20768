#!/usr/bin/env python3
"""Experiment-only: build variants of libcsr5hip whose range kernel (csr5_hot.hip) has one piece of work removed, by
patching a TEMPORARY COPY of the product source (the product file carries no ablation switches).  Results of these
builds are wrong by design; they only answer "what does this piece cost".

    python scripts/experiments/hot_ablation.py            # builds scripts/probes/libcsr5hip_abl_<name>.so for every variant
"""
import os
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SRC = os.path.join(ROOT, "benchmark_spmv_using_csr5_amd", "csrc")

VARIANTS = {
    # name: list of (old, new) replacements in csr5_hot.hip
    "nocold": [("            const unsigned off = cw < 0 ? 0xFFFFFFFFu : (unsigned)cw * (unsigned)sizeof(VT);",
                "            const unsigned off = 0xFFFFFFFFu; (void)cw;")],
    "nogather": [("                    xa[i] = cold_word(a.c[i]);", "                    xa[i] = 0;"),
                 ("                    xa[i] = cold_word(b.c[i]);", "                    xa[i] = 0;")],
    "nostore": [("                if (j > 0 || !open.is_lead)\n                    out[j] = vj;",
                 "                if (vj == (VT)-1.2345e-300)\n                    out[j] = vj;")],
    # (not an ablation: a candidate) partial sums stored with the non-temporal hint
    "ntstore": [("                if (j > 0 || !open.is_lead)\n                    out[j] = vj;",
                 "                if (j > 0 || !open.is_lead)\n                    __builtin_nontemporal_store(vj, out + j);")],
    # (candidates) cache policy of the cold x gathers: gfx940+ aux bits sc0 = 1, nt = 2, sc1 = 16
    "gather_nt": [("__builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 0));\n            else", "__builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 2));\n            else")],
    "gather_sc1": [("__builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 0));\n            else", "__builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 16));\n            else")],
    "gather_sc0sc1": [("__builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 0));\n            else", "__builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 17));\n            else")],
    # conversion: which part of k_row_scan costs its 1.1 ms on R-MAT 24 (wrong format arrays by design)
    "rowscan_noflag": [("        atomicOr(&tile_desc[loc], 1u << (31 - (gbit & 31)));", "        if (gbit == -77) atomicOr(&tile_desc[loc], 1u << (31 - (gbit & 31)));")],
    "rowscan_noempty": [("        atomicOr(&tile_ptr[target], 0x80000000u);", "        if (before == -77) atomicOr(&tile_ptr[target], 0x80000000u);")],
    "rowscan_notileptr": [("        atomicOr(&tile_ptr[t0], (uint32_t)r);\n", "        if (r == -77) atomicOr(&tile_ptr[t0], (uint32_t)r);\n")],
    # (candidates) the combine's partial-sum / row-byte loads with the non-temporal hint (P is dead after the combine)
    "combine_nt": [("                part[q] = P[j];\n                idx[q] = rowidx[j];",
                    "                part[q] = __builtin_nontemporal_load(P + j);\n                idx[q] = __builtin_nontemporal_load(rowidx + j);")],
    "combine_ntP": [("                part[q] = P[j];", "                part[q] = __builtin_nontemporal_load(P + j);")],
    # TIMING ONLY (wrong results: the elements land in the wrong lanes): k_spmv's column / value streams as 16-byte loads (what an
    # in-register transposition of 4 x 4 / 2 x 2 blocks would allow) -- is the one-tile kernel sensitive to its load instruction count?
    "widespmv": [("""#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                c[i] = ct[i * OMEGA];""",
                  """            {
                constexpr int PC = SIGMA % 4 == 0 ? 4 : 2;
                typedef int32_t cp_t __attribute__((ext_vector_type(PC)));
                const cp_t *cp = reinterpret_cast<const cp_t *>(col + (size_t)t * T) + lane;
#pragma unroll
                for (int q = 0; q < SIGMA / PC; q++) {
                    const cp_t w = cp[q * OMEGA];
#pragma unroll
                    for (int e = 0; e < PC; e++)
                        c[q * PC + e] = w[e];
                }
            }"""),
                 ("""#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                v[i] = vt[i * OMEGA];""",
                  """            {
                constexpr int PV = (16 / (int)sizeof(VT)) <= SIGMA && SIGMA % (16 / (int)sizeof(VT)) == 0 ? 16 / (int)sizeof(VT) : 2;
                typedef VT vp_t __attribute__((ext_vector_type(PV)));
                const vp_t *vp = reinterpret_cast<const vp_t *>(val + (size_t)t * T) + lane;
#pragma unroll
                for (int q = 0; q < SIGMA / PV; q++) {
                    const vp_t w = vp[q * OMEGA];
#pragma unroll
                    for (int e = 0; e < PV; e++)
                        v[q * PV + e] = w[e];
                }
            }""")],
    "notable": [("            return __builtin_bit_cast(word_t, hot[cw < 0 ? (unsigned)cw & 0x7FFFFFFFu : 0u]);",
                 "            return (word_t)(unsigned)cw;")],
}


def main():
    names = sys.argv[1:] or list(VARIANTS)
    for name in names:
        tmp = tempfile.mkdtemp(prefix=f"csr5_abl_{name}_")
        for f in os.listdir(SRC):
            if f.endswith((".hip", ".h", ".cpp")):
                shutil.copy(os.path.join(SRC, f), tmp)
        for old, new in VARIANTS[name]:
            # the file a pattern belongs to: csr5_hot.hip unless another source holds it
            for fname in ("csr5_hot.hip", "csr5_format.hip", "csr5_slab.hip", "csr5_spmv.hip"):
                path = os.path.join(tmp, fname)
                text = open(path).read()
                if old in text:
                    open(path, "w").write(text.replace(old, new))
                    break
            else:
                raise SystemExit(f"{name}: pattern not found: {old[:60]}...")
        flags = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-munsafe-fp-atomics", "-Wno-unused-function",
                 f"-I{ROOT}/include", f"-I{tmp}", "-DCSR5_FEW_SIGMAS"]
        objs = []
        procs = []
        for f in ("csr5_format", "csr5_capi", "csr5_ingest", "csr5_slab", "csr5_multi", "csr5_hot"):
            o = os.path.join(tmp, f + ".o")
            objs.append(o)
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, "-c", os.path.join(tmp, f + ".hip"), "-o", o]))
        for t, d in (("f64", "-DCSR5_SPMV_ONLY_F64"), ("f32", "-DCSR5_SPMV_ONLY_F32")):
            o = os.path.join(tmp, f"csr5_spmv_{t}.o")
            objs.append(o)
            procs.append(subprocess.Popen(["/opt/rocm/bin/hipcc", *flags, d, "-c", os.path.join(tmp, "csr5_spmv.hip"), "-o", o]))
        for p in procs:
            if p.wait() != 0:
                raise SystemExit(f"{name}: compile failed")
        out = os.path.join(ROOT, "scripts", "probes", f"libcsr5hip_abl_{name}.so")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-pthread", "-o", out, *objs, "-ldl"])
        shutil.rmtree(tmp)
        print("built", out)


if __name__ == "__main__":
    main()

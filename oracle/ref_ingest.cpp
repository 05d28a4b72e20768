// ref_ingest.cpp -- the REFERENCE's own Matrix Market ingest, captured (TEST INFRASTRUCTURE ONLY).
//
// Compiled only where /root/reference exists (this container), from the reference sources where
// they lie, by oracle/Makefile, into oracle/_ref/libref_ingest.so.  Nothing is copied: the
// reference CLI (CSR5_avx2/main.cpp) is #included by path with two identifiers redirected:
//   main   -> ref_cli_main          so that it can be called as a function;
//   srand  -> a capture hook        main.cpp:283 calls srand(time(NULL)) right after the CSR arrays
//                                   are complete (main.cpp:126-281) and before the file values are
//                                   overwritten by rand()%10 (main.cpp:285-289).  The hook copies
//                                   m, n, nnzA, csrRowPtrA, csrColIdxA, csrValA (the locals of the
//                                   reference's main, by name) and leaves main through an exception,
//                                   so neither the benchmark nor the value overwrite runs.
// The result is the CSR the reference builds from a .mtx file: the pin for oracle/csr5_oracle.c's
// csr5o_mtx_* restatement and for the device COO->CSR kernels.
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <sstream>
#include <vector>

#include "anonymouslib_avx2.h"
#include "mmio.h"

namespace {
struct Captured {
    int m, n, nnz;
    std::vector<int> row_ptr, col_idx;
    std::vector<double> val;
};
Captured g_cap;
struct CaptureDone {};
bool g_stop_after_ingest = true;

void ref_capture(int m, int n, int nnz, const int *rp, const int *ci, const double *v)
{
    if (!g_stop_after_ingest) { // full CLI run (ref_cli_run): behave like the statement we replaced, fixed seed
        srand(20240928u);
        return;
    }
    g_cap.m = m;
    g_cap.n = n;
    g_cap.nnz = nnz;
    g_cap.row_ptr.assign(rp, rp + m + 1);
    g_cap.col_idx.assign(ci, ci + nnz);
    g_cap.val.assign(v, v + nnz);
    throw CaptureDone();
}
} // namespace

#define main ref_cli_main
#define srand(seed) ref_capture(m, n, nnzA, csrRowPtrA, csrColIdxA, csrValA)
#include "main.cpp"
#undef srand
#undef main

// Runs the reference CLI's ingest on `path`.  Returns 0 and fills dims = {m, n, nnz}, or the
// reference's own exit code (-1 cannot open, -2 banner, -3 complex, -4 size line).
extern "C" int ref_ingest_run(const char *path, int *dims)
{
    char prog[] = "spmv";
    std::vector<char> p(path, path + strlen(path) + 1);
    char *argv[] = {prog, p.data(), 0};
    g_cap = Captured();
    // silence the CLI's banners
    std::streambuf *old = std::cout.rdbuf(0);
    int rc = 0;
    bool captured = false;
    try {
        rc = ref_cli_main(2, argv);
    } catch (const CaptureDone &) {
        captured = true;
    }
    std::cout.rdbuf(old);
    std::cout.clear();
    if (!captured) return rc ? rc : -100;
    dims[0] = g_cap.m;
    dims[1] = g_cap.n;
    dims[2] = g_cap.nnz;
    return 0;
}

// Runs the WHOLE reference CLI (ingest, CSR5_avx2 conversion, SpMV loop, self-check) on `path` with a fixed
// rand() seed and returns its stdout in `out` (truncated to cap-1 characters).  BASELINE.json configs[0].
extern "C" int ref_cli_run(const char *path, char *out, int cap)
{
    char prog[] = "spmv";
    std::vector<char> p(path, path + strlen(path) + 1);
    char *argv[] = {prog, p.data(), 0};
    std::ostringstream text;
    std::streambuf *old = std::cout.rdbuf(text.rdbuf());
    g_stop_after_ingest = false;
    int rc = 0;
    try {
        rc = ref_cli_main(2, argv);
    } catch (...) {
        rc = -100;
    }
    g_stop_after_ingest = true;
    std::cout.rdbuf(old);
    const std::string t = text.str();
    const size_t k = t.size() < (size_t)cap - 1 ? t.size() : (size_t)cap - 1;
    memcpy(out, t.data(), k);
    out[k] = 0;
    return rc;
}

extern "C" void ref_ingest_fetch(int *row_ptr, int *col_idx, double *val)
{
    memcpy(row_ptr, g_cap.row_ptr.data(), g_cap.row_ptr.size() * sizeof(int));
    memcpy(col_idx, g_cap.col_idx.data(), g_cap.col_idx.size() * sizeof(int));
    memcpy(val, g_cap.val.data(), g_cap.val.size() * sizeof(double));
}

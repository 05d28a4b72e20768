// ref_spmv.cpp -- thin driver around the REFERENCE's own CSR5_avx2 handle (TEST INFRASTRUCTURE ONLY).
//
// Compiled only where /root/reference exists (this container), from the reference sources where
// they lie, by oracle/Makefile, into oracle/_ref/libref_avx2.so (git-ignored; it travels to the GPU
// box as a prebuilt binary, the sources never do).  It drives `anonymouslibHandle<int, unsigned int,
// double>` exactly as the reference CLI does (CSR5_avx2/main.cpp:18-86): inputCSR, setX,
// setSigma(ANONYMOUSLIB_CSR5_SIGMA), asCSR5, spmv, [50 warm-ups + NUM_RUN timed], destroy.
// Used (a) to generate the y golden vectors, (b) to pin oracle/csr5_oracle.c, and (c) as bench.py's
// cpu_baseline {"kind": "reference"} on the GPU box's host cores.
#include <cstdio>
#include <cstring>
#include <unistd.h>
#include <fcntl.h>

#include "anonymouslib_avx2.h"

namespace {
// asCSR5 prints five lines per call (anonymouslib_avx2.h:119,138,207-210); keep the harness quiet.
struct quiet_stdout {
    int saved;
    quiet_stdout()
    {
        fflush(stdout);
        std::cout.flush();
        saved = dup(1);
        int nul = open("/dev/null", O_WRONLY);
        dup2(nul, 1);
        close(nul);
    }
    ~quiet_stdout()
    {
        fflush(stdout);
        std::cout.flush();
        dup2(saved, 1);
        close(saved);
    }
};
} // namespace

extern "C" int ref_avx2_threads() { return omp_get_max_threads(); }
// the reference takes the ambient OpenMP thread count (csr5_spmv_avx2.h:70); this sets that ambient count
extern "C" void ref_avx2_set_threads(int n) { omp_set_num_threads(n); }
extern "C" int ref_avx2_omega() { return ANONYMOUSLIB_CSR5_OMEGA; }
extern "C" int ref_avx2_sigma() { return ANONYMOUSLIB_CSR5_SIGMA; }

// y is in/out: its initial content is what the reference's first spmv call sees (the CLI zeroes it,
// main.cpp:24).  If runs > 0 the CLI's timing protocol follows (main.cpp:59-74) on a scratch vector.
// Returns the handle's last error code.
extern "C" int ref_avx2_spmv(int m, int n, int nnz, const int *row_ptr, const int *col,
                             const double *val, const double *x, double *y, int warm, int runs,
                             double *ms_per_run, double *convert_ms)
{
    int *rp = (int *)_mm_malloc((size_t)(m + 2) * sizeof(int), 64);
    int *ci = (int *)_mm_malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(int), 64);
    double *va = (double *)_mm_malloc((size_t)(nnz > 0 ? nnz : 1) * sizeof(double), 64);
    double *xx = (double *)_mm_malloc((size_t)(n > 0 ? n : 1) * sizeof(double), 64);
    double *yy = (double *)_mm_malloc((size_t)(m > 0 ? m : 1) * sizeof(double), 64);
    memcpy(rp, row_ptr, (size_t)(m + 1) * sizeof(int));
    rp[m + 1] = 0x7FFFFFFF; // deterministic value for the reference's one-past-the-end read
    memcpy(ci, col, (size_t)nnz * sizeof(int));
    memcpy(va, val, (size_t)nnz * sizeof(double));
    memcpy(xx, x, (size_t)n * sizeof(double));
    memcpy(yy, y, (size_t)m * sizeof(double));

    int err = 0;
    {
        quiet_stdout q;
        anonymouslibHandle<int, unsigned int, double> A(m, n);
        err = A.inputCSR(nnz, rp, ci, va);
        err = A.setX(xx);
        A.setSigma(ANONYMOUSLIB_CSR5_SIGMA);

        anonymouslib_timer conv;
        conv.start();
        err = A.asCSR5();
        if (convert_ms) *convert_ms = conv.stop();

        if (err == 0) err = A.spmv(1.0, yy);

        if (err == 0 && runs > 0) {
            double *yb = (double *)_mm_malloc((size_t)(m > 0 ? m : 1) * sizeof(double), 64);
            memset(yb, 0, (size_t)m * sizeof(double));
            for (int i = 0; i < warm; i++) A.spmv(1.0, yb);
            anonymouslib_timer t;
            t.start();
            for (int i = 0; i < runs; i++) A.spmv(1.0, yb);
            if (ms_per_run) *ms_per_run = t.stop() / (double)runs;
            _mm_free(yb);
        }
        A.destroy();
    }
    memcpy(y, yy, (size_t)m * sizeof(double));
    _mm_free(rp); _mm_free(ci); _mm_free(va); _mm_free(xx); _mm_free(yy);
    return err;
}

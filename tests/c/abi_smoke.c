/* Plain-C host of the C ABI (include/csr5hip.h): what a cgo / JNI / ctypes binding of the reference's
 * anonymouslibHandle would call.  No HIP headers, no C++.  Exit code 0 = y matches a host CSR loop.
 * Built by tests/test_host.py (link check, no GPU) and run by tests/test_gpu_parity.py on the GPU box. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "csr5hip.h"

#define CHECK(call)                                                                                \
    do {                                                                                           \
        int rc_ = (call);                                                                          \
        if (rc_ != CSR5HIP_SUCCESS) {                                                              \
            fprintf(stderr, "%s -> %d: %s\n", #call, rc_, csr5hip_last_error());                   \
            return 2;                                                                              \
        }                                                                                          \
    } while (0)

int main(void)
{
    /* 3000 x 3000, row r holds (r % 7) + (r % 11 == 0 ? 400 : 0) entries: short rows, empty rows, long rows */
    const int m = 3000, n = 3000;
    int *row_ptr = (int *)malloc(sizeof(int) * (m + 1));
    int nnz = 0;
    for (int r = 0; r < m; r++) {
        row_ptr[r] = nnz;
        nnz += r % 7 + (r % 11 == 0 ? 400 : 0);
    }
    row_ptr[m] = nnz;
    int *col = (int *)malloc(sizeof(int) * nnz);
    double *val = (double *)malloc(sizeof(double) * nnz);
    double *x = (double *)malloc(sizeof(double) * n);
    double *y = (double *)malloc(sizeof(double) * m);
    double *y_ref = (double *)calloc(m, sizeof(double));
    unsigned s = 12345u;
    for (int k = 0; k < nnz; k++) {
        s = s * 1664525u + 1013904223u;
        col[k] = (int)((s >> 8) % (unsigned)n);
        val[k] = (double)((s >> 3) % 10u);
    }
    for (int j = 0; j < n; j++)
        x[j] = (double)(j % 10);
    for (int r = 0; r < m; r++)
        for (int k = row_ptr[r]; k < row_ptr[r + 1]; k++)
            y_ref[r] += val[k] * x[col[k]];

    void *d_row_ptr, *d_col, *d_val, *d_x, *d_y;
    CHECK(csr5hip_set_device(0));
    CHECK(csr5hip_malloc(&d_row_ptr, sizeof(int) * (m + 1)));
    CHECK(csr5hip_malloc(&d_col, sizeof(int) * nnz));
    CHECK(csr5hip_malloc(&d_val, sizeof(double) * nnz));
    CHECK(csr5hip_malloc(&d_x, sizeof(double) * n));
    CHECK(csr5hip_malloc(&d_y, sizeof(double) * m));
    CHECK(csr5hip_memcpy_h2d(d_row_ptr, row_ptr, sizeof(int) * (m + 1)));
    CHECK(csr5hip_memcpy_h2d(d_col, col, sizeof(int) * nnz));
    CHECK(csr5hip_memcpy_h2d(d_val, val, sizeof(double) * nnz));
    CHECK(csr5hip_memcpy_h2d(d_x, x, sizeof(double) * n));
    CHECK(csr5hip_memset(d_y, 0, sizeof(double) * m));

    csr5hip_handle A;
    CHECK(csr5hip_create(&A, m, n, CSR5HIP_F64));
    CHECK(csr5hip_input_csr(A, nnz, (int32_t *)d_row_ptr, (int32_t *)d_col, d_val));
    CHECK(csr5hip_set_x(A, d_x));
    CHECK(csr5hip_set_sigma(A, CSR5HIP_AUTO_TUNED_SIGMA));
    if (csr5hip_spmv(A, 1.0, d_y) != CSR5HIP_UNSUPPORTED_CSR_SPMV) /* still CSR: the reference returns -4 */
        return 3;
    CHECK(csr5hip_as_csr5(A));
    csr5hip_info info;
    CHECK(csr5hip_get_info(A, &info));
    CHECK(csr5hip_spmv(A, 1.0, d_y));
    CHECK(csr5hip_synchronize());
    CHECK(csr5hip_memcpy_d2h(y, d_y, sizeof(double) * m));
    CHECK(csr5hip_destroy(A));
    CHECK(csr5hip_free(A));

    int bad = 0;
    for (int r = 0; r < m; r++)
        if (row_ptr[r] != row_ptr[r + 1] && fabs(y[r] - y_ref[r]) > 0.0)
            bad++;
    printf("%s sigma=%d tiles=%d nnz=%d mismatches=%d\n", csr5hip_version(), info.sigma, info.p, nnz, bad);
    csr5hip_device_free(d_row_ptr), csr5hip_device_free(d_col), csr5hip_device_free(d_val);
    csr5hip_device_free(d_x), csr5hip_device_free(d_y);
    return bad ? 1 : 0;
}

/*
 * csr5_oracle.c -- CPU restatement of the reference CSR5 algorithm (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the parity oracle for the HIP path.  It is NOT part of the product: only tests/,
 * __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.  The product
 * (benchmark_spmv_using_csr5_amd/csrc) never links, imports or calls anything in oracle/.
 *
 * Pinning status: PINNED.  Every function here is checked (tests/test_oracle_golden.py) against
 *   (a) golden vectors under tests/golden/ that were produced by compiling the reference's own
 *       CSR5_avx2 sources where they lie (oracle/ref_format.cpp, oracle/ref_spmv.cpp ->
 *       oracle/_ref/*.so, recipe in oracle/Makefile, generator oracle/gen_golden.py), and
 *   (b) the survey's worked known-answer KAT-0.
 *
 * Everything is plain scalar C with omega (lanes per tile) and sigma (elements per lane) as RUNTIME
 * parameters, so one build restates the reference at omega=4/sigma=16 (CSR5_avx2), omega=32 (CSR5_cuda)
 * and omega=64 (ours).  uiT is fixed to 32 bit and iT to int32, as in every reference instantiation
 * (`anonymouslibHandle<int, unsigned int, VALUE_TYPE>`, CSR5_avx2/main.cpp:29).
 *
 * Citations are relative to /root/reference.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifdef _OPENMP
#include <omp.h>
#endif

#define CSR5O_OK 0
#define CSR5O_UNSUPPORTED_OMEGA (-2) /* detail/common.h:15 ANONYMOUSLIB_UNSUPPORTED_CSR5_OMEGA */

/* ---------------------------------------------------------------------------------------------
 * Scalar helpers
 * ------------------------------------------------------------------------------------------- */

/* CSR5_avx2/detail/avx2/utils_avx2.h:23-46 (binary_search_right_boundary_kernel):
 * number of entries of the sorted array that are <= key, i.e. an upper bound. */
static int32_t upper_bound_i32(const int32_t *a, int32_t key, int32_t size)
{
    int32_t lo = 0, hi = size - 1;
    while (hi >= lo) {
        int32_t mid = (int32_t)(((int64_t)hi + lo) / 2);
        if (key >= a[mid])
            lo = mid + 1;
        else
            hi = mid - 1;
    }
    return lo;
}

/* CSR5_avx2/anonymouslib_avx2.h:121-137: derived format parameters.
 * out[0]=bit_y_offset out[1]=bit_scansum_offset out[2]=num_packet out[3]=p */
int csr5o_params(int omega, int sigma, int nnz, int *out)
{
    int base = 2, bit_y = 1, bit_ss = 1;
    while (base < omega * sigma) { base *= 2; bit_y++; }
    base = 2;
    while (base < omega) { base *= 2; bit_ss++; }
    if (bit_y + bit_ss > 31)
        return CSR5O_UNSUPPORTED_OMEGA;
    int bit_all = bit_y + bit_ss + sigma;
    out[0] = bit_y;
    out[1] = bit_ss;
    out[2] = (bit_all + 31) / 32;
    int64_t T = (int64_t)omega * sigma;
    out[3] = (int)((nnz + T - 1) / T);
    return CSR5O_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Step 1: tile_ptr (partition pointer).  CSR5_avx2/detail/avx2/format_avx2.h:7-63.
 * tile_ptr[t] = (#rows r in [0,m] with row_ptr[r] <= min(t*T, nnz)) - 1; bit 31 = "the tile's
 * row range contains an empty row" (only when start != stop).
 *
 * The AVX2 loop at format_avx2.h:48-50 is inclusive of `stop`, which for the LAST tile (stop == m)
 * reads row_ptr[m+1], one past the array.  For every other tile row `stop` can never be empty
 * (it is the last row whose pointer is <= the boundary), so the inclusive and the CUDA variant's
 * exclusive scan (CSR5_cuda/detail/cuda/format_cuda.h:77-89) agree.  We scan [start, min(stop, m-1)],
 * which equals both wherever both are defined.
 * ------------------------------------------------------------------------------------------- */
void csr5o_tile_ptr(int omega, int sigma, int p, int m, int nnz, const int32_t *row_ptr,
                    uint32_t *tile_ptr)
{
    const int64_t T = (int64_t)omega * sigma;
    for (int t = 0; t <= p; t++) {
        int64_t b = (int64_t)t * T;
        int32_t boundary = b > nnz ? nnz : (int32_t)b;
        tile_ptr[t] = (uint32_t)(upper_bound_i32(row_ptr, boundary, m + 1) - 1);
    }
    for (int t = 0; t < p; t++) {
        uint32_t start = tile_ptr[t] & 0x7FFFFFFFu;
        uint32_t stop = tile_ptr[t + 1] & 0x7FFFFFFFu;
        if (start == stop)
            continue;
        int dirty = 0;
        for (uint32_t r = start; r <= stop && r < (uint32_t)m; r++) {
            if (row_ptr[r] == row_ptr[r + 1]) { dirty = 1; break; }
        }
        if (dirty)
            tile_ptr[t] = start | 0x80000000u;
    }
}

/* ---------------------------------------------------------------------------------------------
 * Step 2: tile_desc (partition descriptor) + offset_pointer.
 * CSR5_avx2/detail/avx2/format_avx2.h:88-273.
 *
 * Layout: tile_desc[tile][packet][lane]; packet 0 = y_offset:bit_y | scansum_offset:bit_ss | first
 * (32-B) bit flags MSB first; later packets 32 flags each.  Only tiles 0..p-2 are described (the
 * last tile is processed from CSR, format_avx2.h:98,142); its words stay zero.
 * Returns num_offsets (= offset_ptr[p] after the exclusive scan).
 * ------------------------------------------------------------------------------------------- */
static inline int desc_flag(const uint32_t *desc_tile, int omega, int lane, int i, int bit_all)
{
    /* flag of element i of `lane`: bit index (i + bit_all) counted MSB-first across packets */
    int g = i + bit_all;
    return (desc_tile[(g >> 5) * omega + lane] >> (31 - (g & 31))) & 1u;
}

int csr5o_tile_desc(int omega, int sigma, int p, int m, int bit_y, int bit_ss, int num_packet,
                    const int32_t *row_ptr, const uint32_t *tile_ptr, uint32_t *tile_desc,
                    int32_t *offset_ptr)
{
    const int bit_all = bit_y + bit_ss;
    const int64_t T = (int64_t)omega * sigma;
    (void)m;
    memset(tile_desc, 0, (size_t)p * omega * num_packet * sizeof(uint32_t));
    memset(offset_ptr, 0, (size_t)(p + 1) * sizeof(int32_t));

    /* s1 (format_avx2.h:88-124): one flag per row start that falls inside tiles 0..p-2 */
    for (int t = 0; t < p - 1; t++) {
        int32_t row_start = (int32_t)(tile_ptr[t] & 0x7FFFFFFFu);
        int32_t row_stop = (int32_t)(tile_ptr[t + 1] & 0x7FFFFFFFu);
        for (int32_t r = row_start; r <= row_stop; r++) {
            int32_t ptr = row_ptr[r];
            if (ptr / T != t)
                continue;
            int lane = (int)((ptr / sigma) % omega);
            int g = ptr % sigma + bit_all;
            tile_desc[(size_t)t * omega * num_packet + (g >> 5) * omega + lane] |=
                1u << (31 - (g & 31));
        }
    }

    /* s2 (format_avx2.h:126-236): per-lane segment counts -> y_offset, scansum_offset */
    int *segn_scan = (int *)malloc((size_t)(omega + 1) * sizeof(int));
    int *present = (int *)malloc((size_t)(omega + 1) * sizeof(int));
    for (int t = 0; t < p - 1; t++) {
        uint32_t *d = &tile_desc[(size_t)t * omega * num_packet];
        int with_empty = (tile_ptr[t] >> 31) & 1u;
        uint32_t row_start = tile_ptr[t] & 0x7FFFFFFFu;
        uint32_t row_stop = tile_ptr[t + 1] & 0x7FFFFFFFu;
        if (row_start == row_stop) /* fast-track tile: only raw flags are kept (format_avx2.h:153) */
            continue;
        for (int lane = 0; lane < omega; lane++) {
            int first = desc_flag(d, omega, lane, 0, bit_all) | (lane == 0);
            int start = !first;
            int stop = 0;
            int pres = first;
            for (int i = 1; i < sigma; i++) {
                int f = desc_flag(d, omega, lane, i, bit_all);
                stop += f;
                pres |= f;
            }
            int segn = stop - start + pres;
            segn_scan[lane] = segn > 0 ? segn : 0;
            present[lane] = pres;
        }
        /* exclusive scan over omega+1 entries (utils_avx2.h:72-86 scan_single) */
        int run = 0;
        for (int lane = 0; lane <= omega; lane++) {
            int v = lane < omega ? segn_scan[lane] : 0;
            segn_scan[lane] = run;
            run += v;
        }
        if (with_empty)
            offset_ptr[t] = segn_scan[omega];
        for (int lane = 0; lane < omega; lane++) {
            int y_offset = lane ? segn_scan[lane] - 1 : 0;
            int ss = 0;
            if (present[lane]) {
                int nx = lane + 1;
                while (nx < omega && !present[nx]) { ss++; nx++; }
            }
            d[lane] |= (uint32_t)y_offset << (32 - bit_y);
            d[lane] |= (uint32_t)ss << (32 - bit_all);
        }
    }
    free(segn_scan);
    free(present);

    /* s3 (format_avx2.h:261-264): exclusive scan of offset_ptr[0..p] */
    int32_t run = 0;
    for (int t = 0; t <= p; t++) {
        int32_t v = offset_ptr[t];
        offset_ptr[t] = run;
        run += v;
    }
    return offset_ptr[p];
}

/* ---------------------------------------------------------------------------------------------
 * Step 2b: empty-row offsets.  CSR5_avx2/detail/avx2/format_avx2.h:275-369.
 * For tiles whose tile_ptr carries bit 31: the k-th flag of the tile (lane-major; lane 0's forced
 * first flag excluded) gets offset[offset_ptr[t] + k] = row index relative to row_start+1.
 * Slots that the reference never writes are left untouched (caller pre-fills a sentinel).
 * ------------------------------------------------------------------------------------------- */
void csr5o_desc_offset(int omega, int sigma, int p, int bit_y, int bit_ss, int num_packet,
                       const int32_t *row_ptr, const uint32_t *tile_ptr,
                       const uint32_t *tile_desc, const int32_t *offset_ptr, int32_t *offset)
{
    const int bit_all = bit_y + bit_ss;
    const int64_t T = (int64_t)omega * sigma;
    for (int t = 0; t < p - 1; t++) {
        if (!((tile_ptr[t] >> 31) & 1u))
            continue;
        int32_t row_start = (int32_t)(tile_ptr[t] & 0x7FFFFFFFu);
        int32_t row_stop = (int32_t)(tile_ptr[t + 1] & 0x7FFFFFFFu);
        const uint32_t *d = &tile_desc[(size_t)t * omega * num_packet];
        int32_t base = offset_ptr[t];
        for (int lane = 0; lane < omega; lane++) {
            int y_offset = (int)(d[lane] >> (32 - bit_y));
            for (int i = 0; i < sigma; i++) {
                int f = desc_flag(d, omega, lane, i, bit_all);
                if (i == 0 && lane == 0)
                    continue; /* forced flag of lane 0 is not a store (format_avx2.h:310-312) */
                if (!f)
                    continue;
                int32_t idx = (int32_t)((int64_t)t * T + (int64_t)lane * sigma + i);
                int32_t y_index =
                    upper_bound_i32(&row_ptr[row_start + 1], idx, row_stop - row_start) - 1;
                offset[base + y_offset] = y_index;
                y_offset++;
            }
        }
    }
}

/* ---------------------------------------------------------------------------------------------
 * Step 3: in-place tile transpose.  CSR5_avx2/detail/avx2/format_avx2.h:371-458.
 * Tiles 0..p-2 whose RAW tile_ptr words differ (format_avx2.h:390) move element (lane l, step i)
 * from t*T + l*sigma + i to t*T + i*omega + l (r2c != 0) or back (r2c == 0).
 * ------------------------------------------------------------------------------------------- */
void csr5o_transpose(int omega, int sigma, int nnz, const uint32_t *tile_ptr, void *data,
                     int elem_size, int r2c)
{
    const int64_t T = (int64_t)omega * sigma;
    int num_p = (int)((nnz + T - 1) / T) - 1;
    char *tmp = (char *)malloc((size_t)T * elem_size);
    for (int t = 0; t < num_p; t++) {
        if (tile_ptr[t] == tile_ptr[t + 1])
            continue;
        char *tile = (char *)data + (size_t)t * T * elem_size;
        for (int l = 0; l < omega; l++)
            for (int i = 0; i < sigma; i++) {
                size_t csr_pos = (size_t)l * sigma + i;
                size_t csr5_pos = (size_t)i * omega + l;
                if (r2c)
                    memcpy(tmp + csr5_pos * elem_size, tile + csr_pos * elem_size, elem_size);
                else
                    memcpy(tmp + csr_pos * elem_size, tile + csr5_pos * elem_size, elem_size);
            }
        memcpy(tile, tmp, (size_t)T * elem_size);
    }
    free(tmp);
}

/* ---------------------------------------------------------------------------------------------
 * SpMV.  Tile-level formulation of CSR5_cuda/detail/cuda/csr5_spmv_cuda.h:59-200 (one SIMD group of
 * omega lanes per tile, one calibrator slot per tile) evaluated lane by lane, followed by the
 * calibrate pass (csr5_spmv_cuda.h:313-382) and the CSR tail tile (csr5_spmv_cuda.h:384-419;
 * CSR5_avx2/detail/avx2/csr5_spmv_avx2.h:316-346).
 *
 * y semantics follow CSR5_avx2 (csr5_spmv_avx2.h:42-49, 284-291, 340-344): a row that starts exactly
 * at a tile boundary is OVERWRITTEN by the first carry, all other carries are added; rows without
 * non-zeros in tiles 0..p-2 are never written; every row >= tail_start is written.  So the result
 * does not depend on the caller zeroing y (the CUDA variant needs that).
 * `alpha` is accepted and ignored, as in every reference backend (csr5_spmv_avx2.h:338).
 *
 * The cross-lane step adds lead[l+1 .. l+scansum_offset+1] directly instead of the reference's
 * prefix-sum difference (csr5_spmv_cuda.h:25-38); the two are equal in exact arithmetic.
 * ------------------------------------------------------------------------------------------- */
#define CSR5O_DEFINE_SPMV(NAME, VT)                                                               \
    void NAME(int omega, int sigma, int p, int m, int bit_y, int bit_ss, int num_packet,          \
              const int32_t *row_ptr, const int32_t *col, const VT *val,                          \
              const uint32_t *tile_ptr, const uint32_t *tile_desc, const int32_t *offset_ptr,     \
              const int32_t *offset, VT *calibrator, int tail_start, const VT *x, VT *y)          \
    {                                                                                             \
        const int bit_all = bit_y + bit_ss;                                                       \
        const int64_t T = (int64_t)omega * sigma;                                                 \
        if (p <= 0)                                                                               \
            return;                                                                               \
        _Pragma("omp parallel")                                                                   \
        {                                                                                         \
            VT *lead = (VT *)malloc((size_t)(omega + 2) * sizeof(VT));                            \
            VT *last = (VT *)malloc((size_t)omega * sizeof(VT));                                  \
            VT *firsts = (VT *)malloc((size_t)omega * sizeof(VT));                                \
            int *yoff = (int *)malloc((size_t)omega * sizeof(int));                               \
            int *dir = (int *)malloc((size_t)omega * sizeof(int));                                \
            int *ss = (int *)malloc((size_t)omega * sizeof(int));                                 \
            int *has = (int *)malloc((size_t)omega * sizeof(int));                                \
            _Pragma("omp for schedule(static)")                                                   \
            for (int t = 0; t < p - 1; t++) {                                                     \
                const int32_t *ctile = &col[(size_t)t * T];                                       \
                const VT *vtile = &val[(size_t)t * T];                                            \
                const uint32_t *d = &tile_desc[(size_t)t * omega * num_packet];                   \
                uint32_t rs_raw = tile_ptr[t];                                                    \
                uint32_t row_stop = tile_ptr[t + 1] & 0x7FFFFFFFu;                                \
                if (rs_raw == row_stop) { /* fast track, csr5_spmv_cuda.h:59-90 */                \
                    VT s = 0;                                                                     \
                    for (int64_t k = 0; k < T; k++)                                               \
                        s += vtile[k] * x[ctile[k]];                                              \
                    calibrator[t] = s;                                                            \
                    continue;                                                                     \
                }                                                                                 \
                int empty_rows = (rs_raw >> 31) & 1u;                                             \
                int32_t row_start = (int32_t)(rs_raw & 0x7FFFFFFFu);                              \
                VT *y_local = &y[row_start + 1];                                                  \
                int32_t obase = empty_rows ? offset_ptr[t] : 0;                                   \
                for (int l = 0; l < omega; l++) {                                                 \
                    uint32_t w0 = d[l];                                                           \
                    int y_offset = (int)(w0 >> (32 - bit_y));                                     \
                    ss[l] = (int)((w0 << bit_y) >> (32 - bit_ss));                                \
                    int f0 = desc_flag(d, omega, l, 0, bit_all) | (l == 0);                       \
                    int direct = f0 & (l != 0);                                                   \
                    int stop = 0;                                                                 \
                    VT first_sum = 0;                                                             \
                    VT sum = vtile[l] * x[ctile[l]];                                              \
                    for (int i = 1; i < sigma; i++) {                                             \
                        int f = desc_flag(d, omega, l, i, bit_all);                               \
                        if (f) {                                                                  \
                            if (direct)                                                           \
                                y_local[empty_rows ? offset[obase + y_offset] : y_offset] = sum;  \
                            else                                                                  \
                                first_sum = sum;                                                  \
                            y_offset += direct;                                                   \
                            direct = 1;                                                           \
                            sum = 0;                                                              \
                            stop++;                                                               \
                        }                                                                         \
                        sum += vtile[(size_t)i * omega + l] * x[ctile[(size_t)i * omega + l]];    \
                    }                                                                             \
                    if (!direct)                                                                  \
                        first_sum = sum;                                                          \
                    firsts[l] = first_sum;                                                        \
                    last[l] = sum;                                                                \
                    lead[l] = f0 ? (VT)0 : first_sum; /* start ? first_sum : 0 */                 \
                    has[l] = !(!f0 && stop == 0);     /* start <= stop */                         \
                    yoff[l] = y_offset;                                                           \
                    dir[l] = direct;                                                              \
                }                                                                                 \
                lead[omega] = 0;                                                                  \
                lead[omega + 1] = 0;                                                              \
                for (int l = 0; l < omega; l++) {                                                 \
                    if (has[l]) {                                                                 \
                        VT add = 0;                                                               \
                        for (int j = l + 1; j <= l + ss[l] + 1 && j <= omega; j++)                \
                            add += lead[j];                                                       \
                        last[l] += add;                                                           \
                    }                                                                             \
                    if (dir[l])                                                                   \
                        y_local[empty_rows ? offset[obase + yoff[l]] : yoff[l]] = last[l];        \
                }                                                                                 \
                calibrator[t] = dir[0] ? firsts[0] : last[0];                                     \
            }                                                                                     \
            free(lead); free(last); free(firsts); free(yoff); free(dir); free(ss); free(has);     \
        }                                                                                         \
        /* tail tile, CSR order (csr5_spmv_avx2.h:316-346); first-row partial -> calibrator[p-1] */\
        {                                                                                         \
            const int64_t first_tail = (int64_t)(p - 1) * T;                                      \
            for (int r = tail_start; r < m; r++) {                                                \
                int64_t a = (r == tail_start) ? first_tail : row_ptr[r];                          \
                VT s = 0;                                                                         \
                for (int64_t k = a; k < row_ptr[r + 1]; k++)                                      \
                    s += val[k] * x[col[k]];                                                      \
                if (r == tail_start)                                                              \
                    calibrator[p - 1] = s;                                                        \
                else                                                                              \
                    y[r] = s;                                                                     \
            }                                                                                     \
        }                                                                                         \
        /* calibrate (csr5_spmv_cuda.h:313-382) with CSR5_avx2 overwrite semantics: the first carry \
         * of a row that begins exactly on a tile boundary stores, every other carry adds. */       \
        for (int t = 0; t < p; t++) {                                                             \
            int32_t r = (int32_t)(tile_ptr[t] & 0x7FFFFFFFu);                                     \
            if (r >= m)                                                                           \
                continue; /* only possible for the tail when nothing is left */                   \
            int head = (t == 0) || ((int32_t)(tile_ptr[t - 1] & 0x7FFFFFFFu) != r);               \
            if (head && (int64_t)row_ptr[r] == (int64_t)t * T)                                    \
                y[r] = calibrator[t];                                                             \
            else                                                                                  \
                y[r] += calibrator[t];                                                            \
        }                                                                                         \
    }

CSR5O_DEFINE_SPMV(csr5o_spmv_f64, double)
CSR5O_DEFINE_SPMV(csr5o_spmv_f32, float)

/* Scalar CSR loop, the reference CLI's own check (CSR5_avx2/main.cpp:305-318). */
void csr5o_csr_spmv_f64(int m, const int32_t *row_ptr, const int32_t *col, const double *val,
                        const double *x, double *y)
{
    for (int i = 0; i < m; i++) {
        double s = 0;
        for (int j = row_ptr[i]; j < row_ptr[i + 1]; j++)
            s += x[col[j]] * val[j];
        y[i] = s;
    }
}

void csr5o_csr_spmv_f32(int m, const int32_t *row_ptr, const int32_t *col, const float *val,
                        const float *x, float *y)
{
    for (int i = 0; i < m; i++) {
        float s = 0;
        for (int j = row_ptr[i]; j < row_ptr[i + 1]; j++)
            s += x[col[j]] * val[j];
        y[i] = s;
    }
}

int csr5o_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

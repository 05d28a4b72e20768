/*
 * csr5hip.h -- C ABI of libcsr5hip.so: CSR -> CSR5 conversion and CSR5 SpMV on MI355X (gfx950).
 *
 * This is the drop-in boundary for the reference's `anonymouslibHandle<iT, uiT, vT>` class template
 * (CSR5_cuda/anonymouslib_cuda.h:11-53, CSR5_avx2/anonymouslib_avx2.h:11-52).  Every entry point below
 * names the reference member it replaces.  Semantics follow the CUDA variant of the reference: all
 * matrix/vector pointers are DEVICE pointers that the caller allocates, fills and owns; the handle
 * borrows them and allocates only the CSR5 auxiliary arrays (anonymouslib_cuda.h:142-151,188).
 * `include/anonymouslib_hip.h` re-creates the C++ class template on top of this ABI.
 *
 * Fixed instantiation: iT = int32_t, uiT = uint32_t (the only one the reference ever uses,
 * CSR5_cuda/main.cu:59), vT = double or float chosen at create time.  omega = 64 = one wavefront.
 *
 * No torch / HIP types appear in the signatures: streams are passed as `void*` (a hipStream_t).
 */
#ifndef CSR5HIP_H
#define CSR5HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* detail/common.h:13-22 (same numeric values) */
#define CSR5HIP_SUCCESS                   0
#define CSR5HIP_UNKOWN_FORMAT            (-1)
#define CSR5HIP_UNSUPPORTED_CSR5_OMEGA   (-2)
#define CSR5HIP_CSR_TO_CSR5_FAILED       (-3)
#define CSR5HIP_UNSUPPORTED_CSR_SPMV     (-4)
#define CSR5HIP_UNSUPPORTED_VALUE_TYPE   (-5)
/* additions: the reference aborts the process on runtime failures (checkCudaErrors) */
#define CSR5HIP_HIP_ERROR                (-100)
#define CSR5HIP_INVALID_ARGUMENT         (-101)

#define CSR5HIP_FORMAT_CSR   0
#define CSR5HIP_FORMAT_CSR5  1

#define CSR5HIP_OMEGA             64   /* ANONYMOUSLIB_CSR5_OMEGA (detail/cuda/common_cuda.h:11): lanes per tile */
#define CSR5HIP_AUTO_TUNED_SIGMA (-1)  /* ANONYMOUSLIB_AUTO_TUNED_SIGMA (detail/cuda/common_cuda.h:15) */
#define CSR5HIP_MIN_SIGMA          1
#define CSR5HIP_MAX_SIGMA         32   /* the reference instantiates sigma = 4..32 (csr5_spmv_cuda.h:445-540) */

typedef enum { CSR5HIP_F64 = 0, CSR5HIP_F32 = 1 } csr5hip_value_type;

/* csr5hip_set_option keys */
#define CSR5HIP_OPT_SPMV_MODE   1  /* 0 = two-pass (tiles+tail, then calibrate; same summation order as the
                                          reference's three-kernel scheme)
                                      1 = fused single launch [default] (cut rows are finished by the owning
                                          tile or, for long rows, by the last arriving tile; no second launch).
                                      Both modes are bit-reproducible run to run. */
#define CSR5HIP_OPT_XCD_REMAP   2  /* 1 = contiguous tile ranges per XCD (default), 0 = round robin */
#define CSR5HIP_OPT_X_WINDOW    3  /* fused mode: stage a per-tile slice of x in LDS and gather from it.
                                      0 = off, 1 = auto (default: on when the per-tile 4-KB windows of x
                                      chosen at conversion cover >= 70 % of the non-zeros, sigma >= 16 and,
                                      for fp64, a gather spreads over >= 16 lines of x), 2 = force */

#define CSR5HIP_OPT_LDS_Y       4  /* compact a tile's y segments in LDS and flush them with coalesced
                                      stores: 0 = off, 1 = auto (default: on at <= 20 non-zeros per row),
                                      2 = force (applies while 64*sigma*sizeof(vT) <= 8 KiB) */

#define CSR5HIP_OPT_STREAM_NT   5  /* non-temporal loads for the column/value streams: 0 = off, 1 = auto (default: on
                                      when those streams exceed the 256-MiB Infinity Cache, so that a matrix that
                                      cannot stay cached between SpMVs does not evict x either), 2 = force */

#define CSR5HIP_OPT_COLUMN_SLABS 6  /* column-slab structure for matrices whose x exceeds one XCD's L2 and whose columns
                                      are scattered (power-law graphs): the non-zeros are ALSO kept partitioned by a hash
                                      of the column into S slabs, one slab range per XCD, so the eight L2s cache eight
                                      different parts of x; per-(row, slab) partial sums are added in slab order by a small
                                      second kernel (deterministic, no floating-point atomics).  A kernel-side table like
                                      the x-window: the four CSR5 arrays in csr5hip_info are unaffected.
                                      0 = off, 1 = auto (default), 2/4/8/16/32/64 = that many slabs */
#define CSR5HIP_OPT_SLAB_SHIFT  7  /* log2 of the number of adjacent columns hashed to the same slab (default 4 =
                                      one 128-byte line of fp64 x) */
#define CSR5HIP_OPT_SLAB_HOT    9  /* column slabs only: keep each slab's most used columns of x (16 384 fp64 / 32 768 fp32) in a 128-KB
                                      LDS table of a persistent kernel (power-law inputs put most non-zeros on few columns): 0 = off,
                                      1 = auto (default: on when the table covers >= 25 % of the non-zeros), 2 = force */
#define CSR5HIP_OPT_ZERO_EMPTY_ROWS 8 /* 1 = spmv() also stores 0 into rows without non-zeros (so y is fully defined
                                      without the caller zeroing it -- what a solver that feeds y back as x needs);
                                      0 = reference behaviour (default): empty rows before the tail are left untouched */

#define CSR5HIP_OPT_SLAB_MEMORY_MIB 10 /* upper bound, in MiB, on the device memory the column-slab structure may take
                                      (second copy of column_index / value, partial sums, build temporaries); 0 = none
                                      (default).  A structure that would exceed it -- or whose allocation fails -- is
                                      not built: asCSR5() still succeeds and spmv() runs the plain tile kernel
                                      (csr5hip_info.slab_fallback = 1, csr5hip_last_error() says why).  Only a structure
                                      that was REQUESTED (CSR5HIP_OPT_COLUMN_SLABS >= 2) makes asCSR5() fail, after the
                                      matrix has been put back into CSR. */

#define CSR5HIP_OPT_X_SNAPSHOT 11 /* With an LDS hot table the slab kernel gathers from a private, PERMUTED copy of x (every
                                      slab's table image, then its remaining columns in descending order of use: the part of
                                      x a slab reads is dense and its popular prefix stays in one L2).
                                      0 (default) = the copy is taken by every spmv(): x is read live, exactly like the
                                          reference's texture / __ldg gathers (csr5_spmv_cuda.h:7-23) -- a caller may change
                                          x's contents between spmv() calls without telling the handle;
                                      1 = the copy is taken once per setX() (by the first spmv() after it): the caller
                                          promises that x's CONTENTS do not change until the next setX() -- what the
                                          reference CLI does (CSR5_cuda/main.cu:63 "you only need to do it once!", then
                                          NUM_RUN spmv() calls on the same x).  Call setX() again, with the same pointer,
                                          after writing to x.  Handles without a hot table ignore the option. */
#define CSR5HIP_OPT_NARROW_VALUES 12 /* fp64 matrices with an LDS hot table: 1 = when EVERY value of the matrix is exactly
                                      representable as a (normal) fp32 number -- integer weights, 0/1 adjacency, the
                                      reference CLI's rand() % 10 data (CSR5_cuda/main.cu:229-233) -- the slab kernel
                                      streams the values as fp32 and widens them in registers: 4 bytes less per non-zero,
                                      the same products and sums bit for bit (checked on the device at conversion; any
                                      other matrix keeps its fp64 stream).  0 (default) = off: the value stream is the
                                      8-byte one the roofline's algorithmic bytes count.  csr5hip_info.slab_values_narrowed
                                      says what happened.  +4 bytes per non-zero of device memory. */

/* (option numbers 13 and 14 belonged to the range-walking kernel of round 5, which measured slower on every shape and was taken
   out of the product: scripts/experiments/round5/walk_kernel/) */

#define CSR5HIP_OPT_NARROW_COLUMNS 15 /* x-window kernel: when EVERY tile 0 .. p-2 spans fewer than 32 768 columns (banded / blocked
                                      matrices; any matrix with n <= 32 768) the kernel streams 16-bit column codes -- 15 bits of column
                                      minus the tile's smallest column + the element's row-start flag, two per word, kept in a private
                                      array next to column_index -- instead of the 32-bit column words and the descriptor words: 2.25
                                      bytes less per non-zero, the same gathers, bit-identical results.  Built at conversion when a
                                      windowed kernel is selected (sigma 8, 12, 16, 24 or 32); +2 bytes per non-zero of device memory.
                                      1 = auto (default), 0 = off.  csr5hip_info.narrow_columns says what happened. */

#define CSR5HIP_OPT_DEFER_CARRIES 16 /* fused mode, plain path.  A row cut by a tile boundary is normally finished inside the launch: a tile
                                      re-reads a short spill (<= 64 elements) of its last row from the next tile and owns the row, other
                                      cut rows meet in an arrival protocol (one returning atomic per party at the end of its tile).  On
                                      matrices of many tiles both cost more than they save (the spill loads touch sigma cache lines each
                                      in EVERY tile): there no tile finishes a neighbour's spill, every party parks its partial with a
                                      plain store and a second small launch (k_calibrate, the one rows spanning > 64 tiles already use)
                                      adds them in tile order -- the association of the two-pass mode, bit-identical to it; spmv() stays
                                      one call.  1 = auto (default: by tiles, sigma and average row length), 0 = off, 2 = force.
                                      Takes effect at asCSR5(): set it while the matrix is in CSR form.
                                      csr5hip_info.carries_deferred says what happened. */
#define CSR5HIP_OPT_FLAGGED_COLUMNS 18 /* plain fused kernel at sigma 4 .. 8 (short rows: the auto rule's sigma): the kernel streams a private
                                      copy of the tile-ordered column_index that carries the element's row-start flag in bit 31 instead
                                      of column_index + the descriptor words: 256 bytes and one load instruction less per tile, the same
                                      gathers, bit-identical results; the four reference arrays and column_index stay as they are.
                                      +4 bytes per non-zero of device memory.  1 = auto (default: when the column / value streams exceed
                                      the 256-MiB Infinity Cache -- the saving is bytes, not latency: -2.2 % there, +1..5 % on cache-sized
                                      matrices), 0 = off, 2 = force.  csr5hip_info.flagged_columns says what happened. */
/* (option number 17, CSR5HIP_OPT_CARRY_FINISH of round 6 -- the deferred carries added by trailing workgroups of the tile kernel's own
   launch instead of a second launch -- was parity-green and bit-identical but 2 us SLOWER on nd24k-like (the parties' stores must be
   written through to be seen inside the launch) and was taken out again: scripts/experiments/round6/carry_finish_in_launch/) */

typedef struct csr5hip_handle_s *csr5hip_handle;

/* Host-visible snapshot of the handle's private state (anonymouslib_cuda.h:27-52). */
typedef struct csr5hip_info {
    int format;                 /* _format */
    int m, n, nnz;              /* _m, _n, _nnz */
    int value_type;
    int omega;                  /* 64 */
    int sigma;                  /* _csr5_sigma */
    int bit_y_offset;           /* _bit_y_offset */
    int bit_scansum_offset;     /* _bit_scansum_offset */
    int num_packet;             /* _num_packet */
    int p;                      /* _p */
    int tail_partition_start;   /* _tail_partition_start */
    int num_offsets;            /* _num_offsets */
    const uint32_t *d_tile_ptr;    /* _csr5_partition_pointer                  [p+1]                  */
    const uint32_t *d_tile_desc;   /* _csr5_partition_descriptor               [p*omega*num_packet]   */
    const int32_t  *d_offset_ptr;  /* _csr5_partition_descriptor_offset_pointer [p+1]                 */
    const int32_t  *d_offset;      /* _csr5_partition_descriptor_offset        [num_offsets]          */
    int x_window_tiles;            /* tiles that were given an LDS x-window at conversion (ours)      */
    int x_window_active;           /* 1 if spmv() launches the x-window variant                        */
    int x_window_cover_pct;        /* share of the non-zeros (tiles 0..p-2) inside their tile's window */
    int x_window_lines;            /* mean number of distinct 128-B lines of x under the in-window lanes of one gather */
    double t_malloc_ms, t_tile_ptr_ms, t_tile_desc_ms, t_transpose_ms; /* asCSR5 phase timers (:211-214) */
    int column_slabs;              /* S if spmv() runs on the column-slab structure, else 0 (ours)        */
    int slab_shift;                /* log2(columns per hashing granule)                                   */
    int slab_segments;             /* number of (row, slab) segments = rows of the stacked matrix         */
    int slab_sigma, slab_tiles;    /* geometry of the stacked matrix' CSR5 form                           */
    double t_slab_ms;              /* time asCSR5 spent building the slab structure                       */
    int slab_hot;                  /* 1 if the slab kernel gathers hot columns from an LDS table          */
    int slab_hot_cover_pct;        /* share of the non-zeros whose column has a slot in its slab's table  */
    int slab_fallback;             /* 1 = the column-slab structure was wanted but could not be built (memory): the
                                      plain kernel is in use                                                */
    long long device_bytes;        /* device memory held by the handle (CSR5 arrays, kernel tables, slab structure,
                                      build temporaries it keeps); the caller's CSR, x and y are not counted */
    int slab_x_permuted;           /* 1 = the slab kernel gathers from the permuted copy of x (CSR5HIP_OPT_X_SNAPSHOT)   */
    int slab_cold_entries;         /* entries of that copy behind the table images (columns gathered from memory)        */
    int x_snapshot;                /* CSR5HIP_OPT_X_SNAPSHOT as set                                                      */
    int slab_values_narrowed;      /* 1 = CSR5HIP_OPT_NARROW_VALUES took effect: the slab kernel streams fp32 values       */
    int carries_deferred;          /* 1 = cut rows are finished by a second small launch (CSR5HIP_OPT_DEFER_CARRIES)             */
    int narrow_columns;            /* 1 = the x-window kernel streams 16-bit column codes (CSR5HIP_OPT_NARROW_COLUMNS)            */
    int flagged_columns;           /* 1 = the plain kernel streams column words with the row-start flag in bit 31 (CSR5HIP_OPT_FLAGGED_COLUMNS) */
} csr5hip_info;

/* anonymouslibHandle(m, n) -- anonymouslib_cuda.h:15.  Uses the current HIP device. */
int csr5hip_create(csr5hip_handle *out, int m, int n, int value_type);
/* Releases the C object (the C++ class has no destructor; callers pair destroy()+scope exit). */
int csr5hip_free(csr5hip_handle h);
/* Stream on which conversion and SpMV are enqueued (the reference uses the default stream). */
int csr5hip_set_stream(csr5hip_handle h, void *hip_stream);

/* warmup() -- anonymouslib_cuda.h:55-59 / format_cuda.h:7-19 */
int csr5hip_warmup(csr5hip_handle h);
/* inputCSR(nnz, row_ptr, col_idx, val) -- anonymouslib_cuda.h:61-76; device pointers, borrowed */
int csr5hip_input_csr(csr5hip_handle h, int nnz, int32_t *d_row_ptr, int32_t *d_col_idx, void *d_val);
/* setX(x) -- anonymouslib_cuda.h:222-260; device pointer, borrowed */
int csr5hip_set_x(csr5hip_handle h, const void *d_x);
/* setSigma(sigma | ANONYMOUSLIB_AUTO_TUNED_SIGMA) -- anonymouslib_cuda.h:294-318 */
int csr5hip_set_sigma(csr5hip_handle h, int sigma);
/* asCSR5() -- anonymouslib_cuda.h:105-220: tile_ptr, tile_desc(+offsets), IN-PLACE tile transpose */
int csr5hip_as_csr5(csr5hip_handle h);
/* asCSR() -- anonymouslib_cuda.h:78-102: inverse transpose, drop the CSR5 arrays */
int csr5hip_as_csr(csr5hip_handle h);
/* spmv(alpha, y) -- anonymouslib_cuda.h:262-284.  Asynchronous on the handle's stream.  As in every
 * reference backend `alpha` is accepted and NOT applied (csr5_spmv_cuda.h:22): y = A*x.
 * Every row that owns a non-zero, and every row >= tail_partition_start, is overwritten; other
 * (empty) rows are left untouched.  Unlike the CUDA variant y need not be zeroed by the caller. */
int csr5hip_spmv(csr5hip_handle h, double alpha, void *d_y);
/* `count` back-to-back spmv() calls replayed from one captured hipGraph (the reference CLI's timed
 * loop, CSR5_cuda/main.cu:96-99, without per-launch host cost). */
int csr5hip_spmv_repeat(csr5hip_handle h, double alpha, void *d_y, int count);
/* Cold-cache measurement protocol: `count` SpMVs replayed from ONE hipGraph on hs[0]'s stream, the i-th using handle
 * hs[i % k] and the vector d_ys[i % k].  With k copies of a matrix (each with its own x and y) whose total footprint
 * exceeds the 256-MiB Infinity Cache, every SpMV streams its operands from HBM instead of finding them cached from
 * the previous launch (the reference's timed loop, CSR5_cuda/main.cu:96-99, re-reads one matrix). */
int csr5hip_spmv_rotate(csr5hip_handle *hs, void **d_ys, int k, double alpha, int count);
/* CSR5HIP_OPT_X_SNAPSHOT = 1 only: take the permuted copy of x NOW, on the handle's stream, instead of in front of the first
 * spmv() after setX (so that no spmv() of a timed loop carries it).  A no-op for handles without a hot table, in live-x mode,
 * before asCSR5 / setX, or when the copy is current. */
int csr5hip_snapshot_x(csr5hip_handle h);
/* destroy() -- anonymouslib_cuda.h:286-291 (== asCSR) */
int csr5hip_destroy(csr5hip_handle h);

/* Measured sigma selection (what ANONYMOUSLIB_AUTO_TUNED_SIGMA's per-architecture tables approximate,
 * anonymouslib_cuda.h:297-313): converts with each candidate sigma, times a hipGraph batch of SpMVs into
 * d_y and leaves the matrix in CSR5 with the fastest one.  Call after inputCSR + setX; d_y is overwritten. */
int csr5hip_autotune_sigma(csr5hip_handle h, void *d_y, int *best_sigma, double *best_us);

int csr5hip_set_option(csr5hip_handle h, int option, int value);
int csr5hip_get_info(csr5hip_handle h, csr5hip_info *info);
/* sigma that setSigma(AUTO) would pick for (m, nnz, value_type) on gfx950 */
int csr5hip_auto_sigma(int m, int nnz, int value_type);
const char *csr5hip_last_error(void);
const char *csr5hip_version(void);

/* ---- device shims so that a plain g++ host program (the ./spmv CLI, CSR5_cuda/main.cu:17-117) can
 *      allocate and move the caller-owned arrays without including HIP headers ---- */
int csr5hip_device_count(int *count);
int csr5hip_set_device(int device);
int csr5hip_device_name(int device, char *buf, size_t buflen, double *clock_mhz);
int csr5hip_malloc(void **dptr, size_t bytes);
int csr5hip_device_free(void *dptr);
int csr5hip_memcpy_h2d(void *dst, const void *src, size_t bytes);
int csr5hip_memcpy_d2h(void *dst, const void *src, size_t bytes);
int csr5hip_memset(void *dptr, int value, size_t bytes);
int csr5hip_synchronize(void);
/* event pair on the handle's stream: elapsed device time in ms between the two calls */
int csr5hip_timer_start(csr5hip_handle h);
int csr5hip_timer_stop(csr5hip_handle h, double *ms);

/* ---------------------------------------------------------------------------------------------------
 * Matrix Market ingest and COO -> CSR: the step BEFORE the path.
 * Replaces the serial fscanf loop and the host counting scatter of the reference CLI
 * (CSR5_avx2/main.cpp:126-281, CSR5_cuda/main.cu reads through the same code) with a multi-threaded
 * mmap parser and a device-side stable sort; the resulting CSR is identical, entry for entry, to the
 * one the reference builds (file order inside a row; the mirror of a symmetric off-diagonal follows
 * its original).
 * ------------------------------------------------------------------------------------------------- */
#define CSR5HIP_MTX_CANNOT_OPEN   (-1)   /* CLI exit codes, main.cpp:135-157 */
#define CSR5HIP_MTX_BAD_BANNER    (-2)
#define CSR5HIP_MTX_COMPLEX       (-3)
#define CSR5HIP_MTX_BAD_SIZE      (-4)   /* also: fewer entries in the file than the size line announces */

#define CSR5HIP_FIELD_REAL     0
#define CSR5HIP_FIELD_INTEGER  1
#define CSR5HIP_FIELD_PATTERN  2

/* COO triplets of a .mtx file: 0-based, file order, host memory owned by the library. */
typedef struct csr5hip_mtx {
    int32_t m, n;
    int64_t nz;            /* entries in the file (nnzA_mtx_report, main.cpp:132) */
    int field;             /* CSR5HIP_FIELD_* ; pattern entries get the value 1.0 (main.cpp:198) */
    int symmetric;         /* 1 for a symmetric or hermitian banner (main.cpp:159-163); skew-symmetric is NOT expanded */
    int32_t *row, *col;    /* [nz] */
    double *val;           /* [nz] */
    int threads;           /* parser threads used */
    int fast_path;         /* 1 = parallel line parser, 0 = sequential fscanf-compatible scanner */
    double t_parse_ms;
    int64_t file_bytes;
    int alloc_flags;       /* reserved (0) */
} csr5hip_mtx;

/* Device CSR built from COO; every d_* array is allocated here, release with csr5hip_csr_release. */
typedef struct csr5hip_csr {
    int32_t m, n, nnz;
    int32_t *d_row_ptr;    /* [m+1] */
    int32_t *d_col_idx;    /* [nnz] */
    void *d_val;           /* [nnz] of value_type, or NULL when no values were requested */
    int value_type;
    double t_parse_ms, t_h2d_ms, t_build_ms;   /* filled by csr5hip_mtx_load / csr5hip_coo_to_csr */
} csr5hip_csr;

/* Parse `path` (threads <= 0: one per hardware thread, capped at 64).  Returns 0 or CSR5HIP_MTX_*;
 * CSR5HIP_INVALID_ARGUMENT for an index outside [1,m] x [1,n] (the reference would corrupt memory). */
int csr5hip_mtx_read(const char *path, int threads, csr5hip_mtx *out);
int csr5hip_mtx_release(csr5hip_mtx *mtx);

/* main.cpp:213-275 on the device.  d_row / d_col / d_val: nz COO triplets in file order (d_val may be
 * NULL: structure only, out->d_val = NULL).  symmetric != 0 mirrors every off-diagonal entry right after
 * the original.  Values are converted to value_type (the reference stores VALUE_TYPE, main.cpp:207). */
int csr5hip_coo_to_csr(int32_t m, int32_t n, int64_t nz, const int32_t *d_row, const int32_t *d_col,
                       const double *d_val, int symmetric, int value_type, csr5hip_csr *out);
int csr5hip_csr_release(csr5hip_csr *csr);

/* csr5hip_mtx_read + H2D + csr5hip_coo_to_csr in one call (what `./spmv foo.mtx` does first). */
int csr5hip_mtx_load(const char *path, int threads, int value_type, csr5hip_csr *out);

/* ---------------------------------------------------------------------------------------------------
 * Checkpoint of a converted matrix.  csr5hip_save writes the handle's CSR5
 * state -- row_ptr, column_index / value in tile order and the four format arrays of the reference
 * (_csr5_partition_pointer, _csr5_partition_descriptor, ..._offset_pointer, ..._offset;
 * anonymouslib_cuda.h:27-52) -- to one binary file; csr5hip_load restores it into a NEW handle that is in
 * CSR5 format at once (no conversion pass).  The CSR arrays of the loaded matrix are allocated here and
 * returned in `arrays` (the handle borrows them as after csr5hip_input_csr; release them with
 * csr5hip_csr_release AFTER csr5hip_free).  asCSR / destroy on the loaded handle give back plain CSR order.
 * ------------------------------------------------------------------------------------------------- */
int csr5hip_save(csr5hip_handle h, const char *path);
int csr5hip_load(const char *path, csr5hip_handle *h, csr5hip_csr *arrays);

/* ---------------------------------------------------------------------------------------------------
 * One matrix on the G GPUs of a node (SURVEY.md section 8, row e).  The reference is single-device
 * (CSR5_cuda/main.cu:25-26 `cudaSetDevice(0)`): this is the MI355X addition.  The matrix is cut into G contiguous
 * row blocks balanced by COST = non-zeros + row_weight * rows (split points = upper_bound(cost prefix, g*total/G) - 1,
 * the reference's tile_ptr primitive, utils_cuda.h:25-53; row_weight defaults to 2 -- what a row's y element, pointer
 * and share of the slab combine cost next to a non-zero, measured -- and CSR5HIP_MULTI_OPT_ROW_WEIGHT = 0 gives the
 * plain nnz balance); every block becomes an ordinary handle on its own device and stream; x is
 * replicated ONCE at set_x time by a single RCCL broadcast over xGMI (librccl is opened lazily; device-to-device
 * copies when it is absent or a device is listed twice); y stays sharded; no per-SpMV collective.
 * x CONTRACT of the multi handle: set_x CAPTURES x's contents.  A replica is library-owned -- the caller cannot write to
 * it -- so every shard's kernel-side (permuted) copy of x is taken once, right behind the broadcast on the shard's stream,
 * not by every spmv() (CSR5HIP_OPT_X_SNAPSHOT = 1 is the shards' default; round 6: it was 41 of each R-MAT 24 block's ~180 us).
 * The shards on devices[0] that borrow d_x follow the same contract: after writing to x call set_x again (same pointer).
 * csr5hip_multi_set_option(mh, CSR5HIP_OPT_X_SNAPSHOT, 0) restores live reads for those borrowing shards only.
 * Single host thread, all calls asynchronous per device.  devices[] may list a device several times (several
 * shards on one GPU), which is how a 1-GPU box exercises this path.
 * ------------------------------------------------------------------------------------------------- */
typedef struct csr5hip_multi_s *csr5hip_multi;

typedef struct csr5hip_shard {
    int device;              /* HIP device of the shard */
    int row_lo, row_hi;      /* global rows [row_lo, row_hi) */
    int nnz;
    void *d_y;               /* the shard's result vector on `device` (row_hi - row_lo values) */
    csr5hip_handle handle;   /* the shard's ordinary handle (csr5hip_get_info etc.) */
    int x_broadcast;         /* how the last set_x replicated x: 0 = nothing to replicate, 1 = RCCL broadcast, 2 = copies */
} csr5hip_shard;

int csr5hip_multi_create(csr5hip_multi *out, const int *devices, int G, int m, int n, int value_type);
int csr5hip_multi_free(csr5hip_multi mh);
/* inputCSR for the whole matrix: device pointers on devices[0].  Unlike the single handle the arrays are COPIED into
 * the per-device shards (and rebased); the caller's arrays are not modified by asCSR5 and may be freed afterwards. */
int csr5hip_multi_input_csr(csr5hip_multi mh, int nnz, const int32_t *d_row_ptr, const int32_t *d_col_idx, const void *d_val);
int csr5hip_multi_set_sigma(csr5hip_multi mh, int sigma);
/* csr5hip_set_option on every shard; CSR5HIP_MULTI_OPT_ROW_WEIGHT (0..64, before input_csr) is the handle's own key */
#define CSR5HIP_MULTI_OPT_ROW_WEIGHT 100
#define CSR5HIP_MULTI_DEFAULT_ROW_WEIGHT 2
/* 1 = the shards on devices[0] also read a REPLICA of x that the broadcast fills (root = devices[0] itself) instead of
 * borrowing the caller's vector: with G = 1 the one RCCL collective of this path -- communicator over the device list,
 * grouped ncclBroadcast -- then runs on a single GPU exactly as it does on eight (how the 1-GPU test box proves the
 * RCCL bindings).  Default 0.  Takes effect at the next set_x. */
#define CSR5HIP_MULTI_OPT_OWN_REPLICAS 101
int csr5hip_multi_set_option(csr5hip_multi mh, int option, int value);
int csr5hip_multi_as_csr5(csr5hip_multi mh);
/* setX: d_x on devices[0], n values, borrowed by the shards that live there; ONE broadcast to the other devices, each shard's
 * permuted copy of x (hot-table path) taken right behind it.  Call again after changing x's contents. */
int csr5hip_multi_set_x(csr5hip_multi mh, const void *d_x);
/* spmv on every shard, enqueued on the shards' streams (returns without waiting) */
int csr5hip_multi_spmv(csr5hip_multi mh, double alpha);
int csr5hip_multi_spmv_repeat(csr5hip_multi mh, double alpha, int count);
int csr5hip_multi_synchronize(csr5hip_multi mh);
/* device time between the two calls: the MAXIMUM over the shards' streams */
int csr5hip_multi_timer_start(csr5hip_multi mh);
int csr5hip_multi_timer_stop(csr5hip_multi mh, double *ms_max);
int csr5hip_multi_shard(csr5hip_multi mh, int g, csr5hip_shard *out);
/* correctness checks: the y shards collected into one HOST vector of m values; fill_y presets every y byte */
int csr5hip_multi_gather_y(csr5hip_multi mh, void *h_y);
int csr5hip_multi_fill_y(csr5hip_multi mh, int byte_value);
/* destroy() on every shard (the shards' own copies go back to CSR order) */
int csr5hip_multi_destroy(csr5hip_multi mh);

#ifdef __cplusplus
}
#endif
#endif /* CSR5HIP_H */

// csr5_internal.h -- shared declarations of libcsr5hip.so (gfx950 only; not a public header).
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "csr5hip.h"

namespace csr5 {

constexpr int OMEGA = CSR5HIP_OMEGA;          // one wavefront per tile
#ifndef CSR5_WAVES_PER_BLOCK
#define CSR5_WAVES_PER_BLOCK 1
#endif
// SpMV kernels: one tile per 64-thread workgroup.  Measured on MI355X with several processes per point
// (scripts/experiments/wpb_modes.sh): 1 wave vs 2 vs 4 per workgroup = 338 / 333 / 324 GFLOPS scircuit-like,
// 1 408 / 1 388 / 1 320 nd24k-like, 186 / 183 / 185 webbase-like.
constexpr int WAVES_PER_BLOCK = CSR5_WAVES_PER_BLOCK;
constexpr int BLOCK = OMEGA * WAVES_PER_BLOCK;
// conversion kernels keep 128-thread workgroups (thread-per-item kernels and the per-tile transpose)
constexpr int FMT_WAVES_PER_BLOCK = 2;
constexpr int FMT_BLOCK = OMEGA * FMT_WAVES_PER_BLOCK;
constexpr int BIT_SS = 6;                     // bits of scansum_offset at omega = 64
constexpr uint32_t ROW_MASK = 0x7FFFFFFFu;    // tile_ptr bit 31 = "tile has empty rows"
constexpr int STAMP_WORD = 8;                 // counters[8..15]: wall-clock stamps of the conversion phases
constexpr int COUNTER_WORDS = 16;
constexpr int NUM_XCD = 8;
constexpr int RUN_SERIAL_MAX = 64;            // carry runs up to this many tiles resolve in-kernel; longer ones in k_calibrate
#ifndef CSR5_XWIN_BYTES
#define CSR5_XWIN_BYTES 4096
#endif
constexpr int XWIN_BYTES = CSR5_XWIN_BYTES;              // x-window staged in LDS per wavefront: 1024 fp32 / 512 fp64 columns
constexpr int xwin_elems(int value_size) { return XWIN_BYTES / value_size; }
constexpr int XWIN_MIN_COVER_PCT = 30;        // a tile gets a window if it covers at least this share

// bits of y_offset for a given sigma: smallest b >= 1 with 2^b >= omega*sigma
// (anonymouslib_cuda.h:121-124)
constexpr int bit_y_of(int sigma)
{
    int base = 2, b = 1;
    while (base < OMEGA * sigma) { base *= 2; b++; }
    return b;
}
constexpr int num_packet_of(int sigma) { return (bit_y_of(sigma) + BIT_SS + sigma + 31) / 32; }

// Geometry of one converted matrix; passed by value to every kernel.
struct Geometry {
    int m, n, nnz;
    int sigma;
    int bit_y, bit_all, num_packet;
    int p;            // number of tiles, the last one (p-1) is the CSR tail
    int tile_elems;   // omega * sigma
    int tail_start;   // first row of the tail tile
    int defer;        // 1 = deferred carries (fused mode, decided at conversion): no tile finishes its neighbour's short spill and
                      // every run head at which >= 2 partials meet is marked like a long run -- the parties park, k_calibrate adds
};

// Device arrays of the CSR5 format plus our own launch helpers.
struct DeviceArrays {
    const int32_t *row_ptr;
    int32_t *col;
    void *val;
    uint32_t *tile_ptr;     // [p+1]
    uint32_t *tile_desc;    // [p*omega*num_packet]
    int32_t *offset_ptr;    // [p+1]
    int32_t *offset;        // [num_offsets]
    void *calibrator;       // [p] of vT: leading partial of every tile (tail = slot p-1)
    // fused mode only (not part of the reference format): per-run accumulator and arrival counter
    void *carry_acc;        // [p] of vT, all zero between launches
    uint32_t *carry_cnt;    // [p], all zero between launches
    uint32_t *carry_meta;   // [p] x uint4 per tile: see tile_carry_meta in csr5_format.hip
    uint32_t *tile_hdr;     // [8p] fused kernel: carry_meta[t], carry_meta[t+1].x and the tile_ptr pair in ONE 32-B record
    uint32_t *counters;     // [16] conversion statistics: x-window tiles, covered non-zeros, long runs, gather lines; [4] = workgroups
                            // done; [8..15] = four 64-bit wall-clock stamps, one per conversion phase (k_row_scan, k_tile_desc, ...)
    // column-slab child with an LDS hot table (csr5_slab.hip / k_spmv_range): column words with bit 31 set index the table
    const int32_t *hot_cols;   // [hot_slabs * hot_capacity]
    const int32_t *hot_count;  // [hot_slabs]
    const int32_t *hot_tile0;  // [hot_slabs + 1] first tile of every slab, then [hot_slabs] the slabs each XCD walks
    int hot_slabs, hot_capacity;
    void *range_lead;          // [hot_slabs * HOT_RANGES_PER_SLAB] of vT: leading partial of every wavefront range (csr5_hot.hip)
    uint32_t *range_head;      // [hot_slabs * HOT_RANGES_PER_SLAB + 1] first row of every range (+ the CSR tail), k_range_heads
    // packed column codes of a hot child (k_hot_encode ENC_PACK): 3 bytes per non-zero in the child's CSR order
    const uint16_t *col_lo;
    const uint8_t *col_hi;
    const int32_t *slab_off;   // [hot_slabs + 1] first child element of every slab
    int slab_shift, slab_bits; // the column -> slab map (csr5_slab.hip slab_of)
    // permuted copy of x for the packed codes (k_x_permute): [hot_slabs][hot_capacity] table images, then the cold region
    const void *xperm;
    const int32_t *cold_base;  // [hot_slabs + 1] start of every slab's cold entries inside the cold region
    const int32_t *cold_cols;  // [cold_total] column behind every cold entry
    int cold_total;
    // CSR5HIP_OPT_NARROW_VALUES: the hot child's values (tile order, like val) as fp32 -- every one of them exactly; else nullptr
    const float *val32;
    // narrow column codes of the x-window kernel (csr5_spmv.hip C16): every tile 0 .. p-2 spans fewer than 32 768 columns
    const uint32_t *col16;     // [(p-1) * T / 2] two 16-bit codes per word: elements (2d, lane) | (2d+1, lane) << 16 of tile t at
                               // t * T/2 + d * 64 + lane, code = column - base16[t]; nullptr = not built
    const int32_t *base16;     // [p] smallest column of every tile
    // flagged column words of the plain kernel at small sigma (csr5_spmv.hip C31): the tile-ordered column_index of tiles
    // 0 .. p-2 with the element's row-start flag in bit 31 (a column index is < 2^31); nullptr = not built
    const uint32_t *col31;     // [(p-1) * T]
};

// ---- conversion (csr5_format.hip) ----
hipError_t launch_row_scan(const Geometry &g, const DeviceArrays &d, hipStream_t s);
hipError_t launch_tile_desc(const Geometry &g, const DeviceArrays &d, hipStream_t s);
size_t offset_scan_tmp_bytes(int entries);
hipError_t launch_offset_scan(const Geometry &g, const DeviceArrays &d, void *tmp, size_t tmp_bytes, hipStream_t s);
hipError_t launch_desc_offset(const Geometry &g, const DeviceArrays &d, hipStream_t s);
hipError_t launch_transpose(const Geometry &g, const DeviceArrays &d, int value_type, bool r2c,
                            hipStream_t s);
// a hot child with packed column codes: values only, EVERY tile 0 .. p-2 (the range kernel's fast track sums a one-row tile
// in any order, and the packed codes are read in CSR order)
hipError_t launch_transpose_values(const Geometry &g, const DeviceArrays &d, int value_type, hipStream_t s);
hipError_t launch_tile_tables(const Geometry &g, const DeviceArrays &d, int value_size, uint32_t *host_words, bool export_only,
                              hipStream_t s);
// narrow column codes: codes + per-tile base from the tile-ordered column_index; *wide_tiles += tiles that span >= 32 768 columns
hipError_t launch_col16(const Geometry &g, const DeviceArrays &d, uint32_t *col16, int32_t *base16, uint32_t *wide_tiles,
                        hipStream_t s);
constexpr int COL16_SPAN = 32768; // columns a tile may span for the narrow codes: 15 bits of column, bit 15 = row-start flag
constexpr bool col16_sigma(int sigma) { return sigma == 8 || sigma == 12 || sigma == 16 || sigma == 24 || sigma == 32; }
// flagged column words: column_index | row-start flag << 31, for the sigmas where the descriptor word is a visible share of a tile's
// stream (256 B next to 64 sigma x 12 B: 5.6 % at sigma = 6, 2 % at 16) and one of its few load instructions
hipError_t launch_col31(const Geometry &g, const DeviceArrays &d, uint32_t *col31, hipStream_t s);
constexpr bool col31_sigma(int sigma) { return sigma >= 4 && sigma <= 8; }
hipError_t launch_warmup(hipStream_t s);
// flag[0] |= 1 if row_ptr is not 0 = row_ptr[0] <= ... <= row_ptr[m] = nnz, |= 2 if a column index lies outside [0, n)
hipError_t launch_validate_csr(int m, int n, int nnz, const int32_t *row_ptr, const int32_t *col, uint32_t *flag,
                               hipStream_t s);

// ---- column slabs (csr5_slab.hip) ----
hipError_t slab_scan_tmp_bytes(size_t items, size_t *bytes);
hipError_t slab_select_tmp_bytes(int nnz, size_t *bytes);
hipError_t slab_partition(const Geometry &g, const DeviceArrays &d, int value_type, int S, int bits, int shift,
                          uint32_t *hist, void *scan_tmp, size_t scan_tmp_bytes, int32_t *col2, void *val2,
                          uint32_t *key2, hipStream_t s);
hipError_t slab_count_segments(int nnz, const uint32_t *key2, void *tmp, unsigned int *d_count, hipStream_t s);
hipError_t slab_segments(int nnz, const uint32_t *key2, const void *tmp, int32_t *row_ptr2, unsigned char *rowidx, hipStream_t s);
size_t slab_base_words(int m, int S);
hipError_t slab_tables(int m, int m2, int nnz, int S, int p, const int32_t *row_ptr, int32_t *row_ptr2, const uint32_t *key2,
                       const uint32_t *chunk_start, uint32_t *base, uint32_t *nonempty, hipStream_t s);
hipError_t slab_hot_select(int n, int nnz, int S, int bits, int shift, int capacity, int min_count, int sample_stride,
                           const int32_t *col, uint32_t *cnt, void *hotmap, uint32_t *chist, uint32_t *thr,
                           int32_t *hot_cols, int32_t *hot_count, unsigned long long *covered, hipStream_t s);
hipError_t slab_hot_finish(int S, int p_hist, int p, int T, int nnz, int capacity, const uint32_t *chunk_start,
                           int32_t *hot_count, int32_t *tile0, int32_t *slab_off, hipStream_t s);
size_t slab_cold_words(int n, int S, int bits, int shift);
hipError_t slab_cold_sort_tmp_bytes(size_t total, int slab_bits, size_t *bytes);
hipError_t slab_hot_pack(int n, int nnz, int T, int p, int S, int bits, int shift, const int32_t *slab_off, const void *hotmap,
                         const uint32_t *cnt, const uint32_t *key2, int32_t *col2, uint16_t *col_lo, uint8_t *col_hi, uint8_t *ref, uint32_t *rank_of,
                         uint32_t *keys, uint32_t *keys_sorted, uint32_t *src_sorted, void *sort_tmp, size_t sort_tmp_bytes,
                         int32_t *cold_base, int32_t *cold_cols, hipStream_t s);
size_t slab_local_columns(int n, int bits, int shift); // columns of one slab (slab-local ids 0 .. this - 1)
size_t slab_hotmap_bytes(int n, int S, int bits, int shift); // slab-major bitmap of the hot columns + group prefixes
int slab_hot_buckets();
hipError_t launch_slab_combine(int m, int tail_start, int zero_empty, int S, int value_type, const uint32_t *base,
                               const unsigned char *rowidx, const uint32_t *nonempty, const void *P, int segments, void *y,
                               hipStream_t s);

// ---- SpMV (csr5_spmv.hip) ----
struct SpmvOptions {
    int mode;        // CSR5HIP_OPT_SPMV_MODE
    int xcd_remap;   // CSR5HIP_OPT_XCD_REMAP
    int x_window;    // resolved: 1 = launch the LDS x-window variant of the fused kernel
    int lds_y;       // resolved: 1 = compact y segments through LDS before storing them
    int stream_nt;   // resolved: 1 = column/value streams use non-temporal loads
    int long_runs;   // resolved: the matrix has rows spanning > RUN_SERIAL_MAX tiles (fused mode adds k_calibrate)
    int hot;         // resolved: column-slab child whose columns are hot-encoded: persistent k_spmv_range + k_range_finish
    int col16;       // resolved: 1 = the x-window kernel streams 16-bit column codes (2 bytes per non-zero less)
    int col31;       // resolved: 1 = the plain kernel streams the flagged column words (no descriptor load)
};
// deferred carries, auto rule (measured break-evens, scripts/experiments/round5/defer_ab.py): what deferral saves grows with the
// number of tiles -- per tile the two scattered spill loads (sigma cache lines each) and, where rows are too long for short-spill
// ownership, the hand-shake atomics -- what it costs is one small launch (2-2.5 us)
constexpr int DEFER_AUTO_LONG_ROW = 128;          // average non-zeros per row from which most cut rows hand-shake ...
constexpr int DEFER_AUTO_MIN_TILES_LONG = 3000;   // ... and deferral pays from this many tiles on (nd24k-like: 3 741 tiles -2.4 us)
constexpr int DEFER_AUTO_MIN_TILE_SIGMA = 500000; // shorter rows: from tiles x sigma >= this (27 per row, sigma 16: loses 2 us at 21 k
                                                  // tiles, wins 8 at 53 k; 81 per row even at 24 k; R-MAT 20 at 16 k would win 5 %)
constexpr int HOT_AUTO_MIN_COVER_PCT = 25;                  // auto rule of the LDS hot table: share of the non-zeros it must cover ...
constexpr long long HOT_AUTO_MIN_COVERED_BEYOND = 5000000;  // ... and non-zeros covered beyond that share (csr5_capi.hip build_slabs_impl)
constexpr int HOT_LDS_BYTES = 128 * 1024;  // upper bound of the LDS table of hot x entries per workgroup (k_spmv_range)
constexpr int HOT_WAVE_LDS = 4096;         // per-wavefront y-compaction region of k_spmv_range
// Wavefronts of the persistent workgroup (one workgroup per CU).  The kernel is bound by the L1's outstanding requests,
// not by occupancy, so every y-compaction region given back to the table buys coverage: same-call A/B on R-MAT 24 / 22
// (profiles/r04_probes.txt) 16 wavefronts + 12 288 fp64 slots 1 197 / 260 us, 12 + 14 336: 1 181 / 253, 8 + 16 384:
// 1 178 / 253, 6 + 17 408: 1 200 / 266, 4 + 18 432: 1 304 / 300.
#ifndef CSR5_HOT_WAVES
#define CSR5_HOT_WAVES 8
#endif
constexpr int HOT_WAVES = CSR5_HOT_WAVES;
constexpr int HOT_WGS_PER_XCD = 32;        // one workgroup per CU
constexpr int HOT_RANGES_PER_SLAB = HOT_WGS_PER_XCD * HOT_WAVES; // every wavefront of the slab's XCD owns one tile range
#ifndef CSR5_HOT_DEPTH
#define CSR5_HOT_DEPTH 2                   // tiles whose streams are in flight per wavefront (1 or 2)
#endif
// what the persistent kernel needs to know about the slabs (device arrays built by csr5_slab.hip)
struct HotParams {
    int slabs, rounds, capacity;     // S, S / 8, table capacity in elements
    const int32_t *count;            // [slabs] slots in use
    const int32_t *tile0;            // [slabs + 1] first tile owned by each slab (tile0[S] = p - 1); behind it [slabs]: the slabs of
                                     // XCD 0 (one per round), of XCD 1, ... (dealt by size at conversion)
    // packed column codes (DeviceArrays::col_lo ...)
    const uint16_t *col_lo;
    const uint8_t *col_hi;
    const int32_t *slab_off;
    const void *xp;                  // permuted copy of x (DeviceArrays::xperm)
    const int32_t *cold_base;
    int cold_total;
};
// sigma of a hot slab child: a tile's values fill the y-compaction region exactly (8 for fp64, 16 for fp32: multiples of four,
// so a lane's 3-byte column codes are whole dwords), whatever the parent's sigma.  Measured: the hot kernel is flat in sigma
// between 8 and 16 (R-MAT 22 320 / 324 / 329 us at 8 / 12 / 16) and 8 % slower at 4; a child that inherited a parent's
// sigma of 6 or 7 (row blocks of short rows) used to fall off the packed-code path.
constexpr int hot_child_sigma(int value_size) { return HOT_WAVE_LDS / (OMEGA * value_size); }
hipError_t launch_spmv(const Geometry &g, const DeviceArrays &d, int value_type, const void *x,
                       void *y, const SpmvOptions &opt, hipStream_t s);
// csr5_hot.hip: the slab child's SpMV when its column words are hot-encoded (persistent range kernel + finish)
hipError_t launch_spmv_hot(const Geometry &g, const DeviceArrays &d, int value_type, const void *x, void *y,
                           const SpmvOptions &opt, hipStream_t s);
hipError_t prepare_spmv_hot(const Geometry &g, const DeviceArrays &d, int value_type, const SpmvOptions &opt);
hipError_t launch_fp32_exact(const double *v, size_t n, unsigned *flag, hipStream_t s);
hipError_t launch_narrow(const double *v, size_t n, float *o, int tile_elems, int transposed_tiles, hipStream_t s);
hipError_t launch_range_heads(const Geometry &g, const DeviceArrays &d, hipStream_t s);
// the permuted copy of x behind the packed codes of a hot child: xperm[i] = x[hot_cols[i]] for the table images,
// xperm[slabs * capacity + i] = x[cold_cols[i]] for the cold region
hipError_t launch_x_permute(const Geometry &g, const DeviceArrays &d, int value_type, const void *x, hipStream_t s);

} // namespace csr5

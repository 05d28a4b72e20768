// csr5_format.hip -- CSR -> CSR5 conversion kernels for gfx950 (wave64), written from scratch.
//
// What is computed is fixed by the reference (bit-exact tile_ptr / tile_desc / offsets, checked against
// goldens produced by the reference's own format code at omega = 64); HOW is ours:
//
//   reference step (CSR5_cuda/detail/cuda/format_cuda.h)           here
//   ------------------------------------------------------------   --------------------------------------
//   K1 generate_partition_pointer_s1  :21-41   thread / tile,      k_row_scan        thread / ROW: a row owns the tile
//      one bisection of row_ptr per tile                             boundaries inside its pointer range; no search
//   K2 generate_partition_pointer_s2  :43-95   block / tile,       (same kernel)     an empty row with pointer e marks
//      loops over the tile's rows                                    tile (e-1)/T; no loop
//   K4 generate_partition_descriptor_s1 :129-159 thread / row,     k_tile_desc       lane / sigma elements: the distinct row
//      one atomicOr per row                                          pointers inside them, found in the tile's slice of row_ptr
//   K5 generate_partition_descriptor_s2 :161-267 warp / tile,      (same kernel)     wave / tile: popcounts, DPP-free
//      LDS scan + serial look-ahead loop                             shfl scan, ballot + ctz for scansum_offset
//   K6 generate_partition_descriptor_s3 :269-300 1 block           rocprim device scan (k_offset_scan_small: 1 block, <= 16 k tiles)
//   K7 generate_partition_descriptor_offset :362-523               k_desc_offset     wave / flagged tile
//   K8 aosoa_transpose_kernel_smem      :525-744 block / tile,     k_transpose       block / tile, col_idx AND value
//      two launches (col, val), 29 sigma instantiations              in one launch, runtime sigma, padded LDS
#include "csr5_internal.h"

#include <rocprim/device/device_scan.hpp>

namespace csr5 {

// number of entries of a[0..size) that are <= key (the reference's
// binary_search_right_boundary_kernel, utils_cuda.h:25-53, restated as a half-open bisection)
__device__ __forceinline__ int upper_bound(const int32_t *__restrict__ a, int key, int size)
{
    int lo = 0, hi = size;
    while (lo < hi) {
        int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (a[mid] <= key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// Phase boundaries of the conversion (the four "CSR->CSR5 ... time" lines the reference prints,
// anonymouslib_cuda.h:211-214) without events on the stream: the first thread of the first kernel of every phase
// stores the constant-rate wall clock (s_memrealtime) into counters[8 + 2*slot]; the host turns the differences into
// milliseconds with hipDeviceAttributeWallClockRate.  An event record costs 2-4 us of stream time each, four of them
// were a sixth of a small matrix's conversion.
__device__ __forceinline__ void stamp_phase(uint32_t *__restrict__ counters, int slot)
{
    if (counters && blockIdx.x == 0 && threadIdx.x == 0)
        *reinterpret_cast<unsigned long long *>(counters + STAMP_WORD + 2 * slot) = wall_clock64();
}

// ---------------------------------------------------------------------------------------------
// K1 + K2 fused, one thread per row r < m, reading row_ptr once (tile_ptr starts zeroed; every store is an atomicOr of
// disjoint bits, so the parts need no order among themselves).  (Until round 3 the bit flags -- K4 -- were scattered from
// here as well, one atomicOr per non-empty row: 0.8-1.0 ms of the 1.1 ms this pass took on R-MAT 24 and on its 40 M-row
// slab child; k_tile_desc now derives a lane's flags from the tile's slice of row_ptr without atomics.)
//  * K1: tile_ptr[t] = last row r in [0, m] with row_ptr[r] <= min(t*T, nnz).  For t*T < nnz that is the one
//    NON-EMPTY row whose range [row_ptr[r], row_ptr[r+1]) holds t*T, so every row writes the boundaries inside its own
//    range (none for most rows, thousands for a hub row: the wavefront shares those) and tile_ptr[p] = m.  The
//    reference bisects row_ptr once per tile (format_cuda.h:21-41): p dependent 17-step chains, 7 us on a 171 k-row
//    matrix where this pass costs nothing extra;
//  * K2: an EMPTY row with e > 0 lies in the row range [tile_ptr[t], tile_ptr[t+1]) of exactly one
//    tile, t = (e-1)/T  (tile_ptr[t] is the last row with pointer <= t*T, so r > tile_ptr[t] iff
//    e > t*T, and r < tile_ptr[t+1] iff e <= (t+1)*T); leading empty rows (e == 0) precede every
//    tile.  That replaces the reference's per-tile row loop by one store per run of empty rows.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(FMT_BLOCK) k_row_scan(Geometry g, const int32_t *__restrict__ row_ptr,
                                                    uint32_t *__restrict__ tile_ptr,
                                                    uint32_t *__restrict__ tile_desc, uint32_t *__restrict__ counters)
{
    stamp_phase(counters, 0);
    const int r = blockIdx.x * FMT_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & (OMEGA - 1);
    const bool live = r < g.m;
    const int e = live ? row_ptr[r] : 0;
    const int e1 = live ? row_ptr[r + 1] : 0;
    const int T = g.tile_elems;
    if (r == 0)
        atomicOr(&tile_ptr[g.p], (uint32_t)g.m);
    // boundaries t*T in [e, e1): t0 = ceil(e / T) .. t1 = (e1 - 1) / T  (t1 <= p-1 because e1 <= nnz)
    const int t0 = e / T + (e % T != 0);
    const int t1 = e1 > e ? (e1 - 1) / T : t0 - 1;
    const int owned = t1 - t0 + 1;
    if (owned > 0 && owned <= 2) {
        atomicOr(&tile_ptr[t0], (uint32_t)r);
        if (owned == 2)
            atomicOr(&tile_ptr[t1], (uint32_t)r);
    }
    for (unsigned long long wide = __ballot(owned > 2); wide; wide &= wide - 1) {
        const int src = __builtin_ctzll(wide);
        const int a = __shfl(t0, src, OMEGA), b = __shfl(t1, src, OMEGA), row = __shfl(r, src, OMEGA);
        for (int t = a + lane; t <= b; t += OMEGA)
            atomicOr(&tile_ptr[t], (uint32_t)row);
    }
    // Empty rows come in runs with the same pointer (R-MAT: half of all rows) and several runs fall into one tile: only
    // the first lane of the wavefront that names a tile marks it (the targets rise with the row, so that is the lane whose
    // target exceeds the running maximum; one atomic per RUN was 4 M atomics = 0.28 ms of this pass on R-MAT 24).
    const int target = (live && e == e1 && e > 0) ? (e - 1) / T : -1;
    int seen = target;
#pragma unroll
    for (int d = 1; d < OMEGA; d <<= 1) {
        const int o = __shfl_up(seen, d, OMEGA);
        if (lane >= d)
            seen = o > seen ? o : seen;
    }
    const int before = __shfl_up(seen, 1, OMEGA);
    if (target >= 0 && (lane == 0 || target > before))
        atomicOr(&tile_ptr[target], 0x80000000u);
}

// ---------------------------------------------------------------------------------------------
// K5: one wavefront per tile t < p-1.  Lane l owns elements [l*sigma, (l+1)*sigma) of the tile.
//   segn  = number of segments (rows) that start in the lane
//   y_off = exclusive wave prefix of segn, minus one for lanes > 0 (index into y_local)
//   ss    = number of directly following lanes without any flag (scansum_offset)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(FMT_BLOCK) k_tile_desc(Geometry g, const int32_t *__restrict__ row_ptr,
                                                     const uint32_t *__restrict__ tile_ptr,
                                                     uint32_t *__restrict__ tile_desc,
                                                     int32_t *__restrict__ offset_ptr, uint32_t *__restrict__ counters)
{
    stamp_phase(counters, 1);
    const int lane = threadIdx.x & (OMEGA - 1);
    const int t = blockIdx.x * FMT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (t >= g.p - 1)
        return;
    const uint32_t raw = tile_ptr[t];
    const uint32_t row_start = raw & ROW_MASK;
    const uint32_t row_stop = tile_ptr[t + 1] & ROW_MASK;

    // K4 (format_cuda.h:129-160 scatters one bit per row): the flags of this lane's sigma elements [lo, lo + sigma) =
    // the DISTINCT row pointers inside that range.  Rows that start in the tile lie in [row_start, row_stop]; the lane
    // bisects that slice (a few dozen entries, shared by the wavefront in L1) for its first row and hops from one distinct
    // pointer to the next (runs of empty rows share a pointer).  Tiles 0..p-2 only: the last tile is processed from CSR and
    // its descriptor stays zero, as in CSR5_avx2 format_avx2.h:98.
    uint32_t w0 = 0, w1 = 0;
    {
        const int lo = t * g.tile_elems + lane * g.sigma, hi = lo + g.sigma;
        // the tile's slice of row_ptr, staged in LDS when it is short (the usual case: ~T / mean row length entries)
        constexpr int SLICE = 512;
        __shared__ int32_t slice_all[FMT_WAVES_PER_BLOCK][SLICE];
        int32_t *mine = slice_all[threadIdx.x >> 6];
        const int nrows = (int)(row_stop - row_start) + 1;
        const bool staged = nrows <= SLICE;
        if (staged) {
            for (int i = lane; i < nrows; i += OMEGA)
                mine[i] = row_ptr[row_start + i];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
        auto rp_at = [&](int r) -> int { return staged ? mine[r - (int)row_start] : row_ptr[r]; };
        // first row in [row_start, row_stop] whose pointer is >= `key` (row_stop + 1 if none)
        auto first_at_least = [&](int from, int key) {
            int a = from, b = (int)row_stop + 1;
            while (a < b) {
                const int mid = (int)(((unsigned)a + (unsigned)b) >> 1);
                if (rp_at(mid) < key)
                    a = mid + 1;
                else
                    b = mid;
            }
            return a;
        };
        int r = first_at_least((int)row_start, lo);
        while (r <= (int)row_stop) {
            const int e = rp_at(r);
            if (e >= hi)
                break;
            const int gbit = e - lo + g.bit_all;
            if (gbit < 32)
                w0 |= 1u << (31 - gbit);
            else
                w1 |= 1u << (63 - gbit);
            r = first_at_least(r + 1, e + 1);
        }
    }
    uint32_t *d = tile_desc + (size_t)t * OMEGA * g.num_packet;
    if (g.num_packet > 1)
        d[OMEGA + lane] = w1;
    if (row_start == row_stop) { // fast-track tile keeps only its raw flags (format_cuda.h:187-189)
        d[lane] = w0;
        return;
    }

    // all flags of the lane, MSB first (element i -> bit 31-i); sigma <= 32 so one word holds them
    uint32_t flags = w0 << g.bit_all;
    if (g.num_packet > 1)
        flags |= w1 >> (32 - g.bit_all);

    const int f0 = (int)(flags >> 31) | (lane == 0);
    const int stop = __popc(flags & 0x7FFFFFFFu);
    const int present = f0 | (stop > 0);
    int segn = stop - !f0 + present;
    segn = segn > 0 ? segn : 0;

    int incl = segn; // inclusive wave scan
#pragma unroll
    for (int d_ = 1; d_ < OMEGA; d_ <<= 1) {
        int v = __shfl_up(incl, d_, OMEGA);
        if (lane >= d_)
            incl += v;
    }
    if ((raw >> 31) && lane == OMEGA - 1)
        offset_ptr[t] = incl; // segments of this tile; scanned into offsets by launch_offset_scan
    const int y_off = lane ? incl - segn - 1 : 0;

    const unsigned long long pmask = __ballot(present);
    int ss = 0;
    if (present) {
        const unsigned long long rest = lane == OMEGA - 1 ? 0ull : pmask >> (lane + 1);
        ss = rest ? __builtin_ctzll(rest) : OMEGA - 1 - lane;
    }
    w0 |= (uint32_t)y_off << (32 - g.bit_y);
    w0 |= (uint32_t)ss << (32 - g.bit_all);
    d[lane] = w0;
}

// ---------------------------------------------------------------------------------------------
// K6: exclusive scan of offset_ptr[0..p] in place (format_cuda.h:269-300 runs it in ONE 256-thread block; at
// p = 262 k that single workgroup took ~0.3 ms here).  Device-wide decoupled look-back scan (rocprim).
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// K7: tiles whose tile_ptr carries bit 31: the k-th store slot of the tile gets the row index
// (relative to row_start+1) of the segment that starts there.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(FMT_BLOCK) k_desc_offset(Geometry g, const int32_t *__restrict__ row_ptr,
                                                       const uint32_t *__restrict__ tile_ptr,
                                                       const uint32_t *__restrict__ tile_desc,
                                                       const int32_t *__restrict__ offset_ptr,
                                                       int32_t *__restrict__ offset)
{
    const int lane = threadIdx.x & (OMEGA - 1);
    const int t = blockIdx.x * FMT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (t >= g.p - 1)
        return;
    const uint32_t raw = tile_ptr[t];
    if (!(raw >> 31))
        return;
    const int row_start = (int)(raw & ROW_MASK);
    const int row_stop = (int)(tile_ptr[t + 1] & ROW_MASK);
    const uint32_t *d = tile_desc + (size_t)t * OMEGA * g.num_packet;
    const uint32_t w0 = d[lane];
    uint32_t flags = w0 << g.bit_all;
    if (g.num_packet > 1)
        flags |= d[OMEGA + lane] >> (32 - g.bit_all);
    if (lane == 0)
        flags &= 0x7FFFFFFFu; // lane 0's first element never owns a store slot (format_cuda.h:391-399)
    int slot = offset_ptr[t] + (int)(w0 >> (32 - g.bit_y));
    const int32_t *rows = row_ptr + row_start + 1;
    const int nrows = row_stop - row_start;
    const int elem0 = t * g.tile_elems + lane * g.sigma;
    while (flags) {
        const int i = __builtin_clz(flags);
        flags &= ~(0x80000000u >> i);
        offset[slot++] = upper_bound(rows, elem0 + i, nrows) - 1;
    }
}

// ---------------------------------------------------------------------------------------------
// K8: in-place sigma x omega tile transpose of column_index AND value, one workgroup per tile.
// r2c: element (lane l, step i) moves from l*sigma+i to i*omega+l; !r2c is the inverse.
// LDS rows are padded by one element so both passes are bank-conflict free.
// ---------------------------------------------------------------------------------------------
__host__ __device__ inline int transpose_pitch(int sigma) { return OMEGA + (OMEGA / sigma > 0 ? OMEGA / sigma : 1); }

template <typename VT>
__global__ void __launch_bounds__(FMT_BLOCK) k_transpose(Geometry g, const uint32_t *__restrict__ tile_ptr,
                                                     int32_t *__restrict__ col, VT *__restrict__ val,
                                                     int r2c, int tiles_per_block,
                                                     uint32_t *__restrict__ counters)
{
    stamp_phase(counters, 2);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // A workgroup moves `tiles_per_block` consecutive tiles: the bytes a compute unit has in flight are what bounds this
    // pass, and eight resident workgroups with one 6-KB tile each (sigma = 8, the slab child) kept it at 6 TB/s.
    const int t0 = blockIdx.x * tiles_per_block;
    const int T = g.tile_elems;
    const int sigma = g.sigma;
    // row pitch of the staging area: 64 consecutive CSR ranks are sigma steps x 64/sigma lanes, so a pitch of
    // 64 + 64/sigma spreads them over all banks (two per bank, the minimum); 65 put them on sigma + 64/sigma - 1 banks
    const int pitch = transpose_pitch(sigma);
    const size_t per_tile = (size_t)sigma * pitch;
    VT *sv = reinterpret_cast<VT *>(smem);
    int32_t *sc = reinterpret_cast<int32_t *>(smem + per_tile * tiles_per_block * sizeof(VT));
    __shared__ int moved[16]; // (tiles_per_block <= 16)
    if ((int)threadIdx.x < tiles_per_block) {
        const int t = t0 + (int)threadIdx.x;
        // fast-track tiles are not transposed; the test is on the RAW words (format_cuda.h:540)
        moved[threadIdx.x] = t < g.p - 1 && tile_ptr[t] != tile_ptr[t + 1];
    }
    __syncthreads();
    const size_t base = (size_t)t0 * T;
    const int total = tiles_per_block * T;
    for (int e = threadIdx.x; e < total; e += FMT_BLOCK) {
        const int tt = e / T, idx = e - tt * T;
        if (!moved[tt])
            continue;
        int i, l;
        if (r2c) { i = idx % sigma; l = idx / sigma; }
        else     { l = idx & (OMEGA - 1); i = idx >> 6; }
        sc[tt * per_tile + i * pitch + l] = col[base + e];
        sv[tt * per_tile + i * pitch + l] = val[base + e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += FMT_BLOCK) {
        const int tt = e / T, idx = e - tt * T;
        if (!moved[tt])
            continue;
        int i, l;
        if (r2c) { l = idx & (OMEGA - 1); i = idx >> 6; }
        else     { i = idx % sigma; l = idx / sigma; }
        col[base + e] = sc[tt * per_tile + i * pitch + l];
        val[base + e] = sv[tt * per_tile + i * pitch + l];
    }
}

// ---------------------------------------------------------------------------------------------
// Fused-SpMV helper (not part of the reference format): who resolves the partial sums that a tile
// boundary cuts.  One uint4 per tile t (tail = tile p-1):
//   .x bits 0..23 : fallback protocol only -- number of arrivals expected at slot t (t = run head)
//      bit 26 LONG_RUN   : (run heads) the run has more than RUN_SERIAL_MAX tiles: every party only parks its
//                          partial (plain store) and a k_calibrate launch after the tile kernel sums them
//      bit 27 HAS_CLOSING: (run heads) the closing segment of tile t-1 is one of the arrivals
//      bit 28 LEAD_SKIP  : the leading partial of tile t is recomputed by tile t-1; do not emit it
//      bit 29 CLOSE_LOCAL: the closing segment of tile t continues for .z <= 64 elements into tile
//                          t+1 and ends there; tile t reads those elements itself and stores y
//      bit 30 CLOSE_CARRY: the closing segment of tile t continues into tile t+1
//   .y : run head of tile t = slot at which the leading partial of tile t arrives (fallback protocol)
//   .z : number of elements of tile t+1 that belong to the closing row of tile t (CLOSE_LOCAL)
// A "run" is a maximal range of tiles h..e whose first element lies in the same row r.  If r starts
// inside tile h-1, ends inside tile h and spills at most 64 elements ("short spill", the common case
// for short rows), tile h-1 owns y[r] outright and nothing is communicated.  Otherwise every partial
// of r arrives at slot h and the last arriver stores y[r] (csr5_spmv.hip carry_arrive).
// ---------------------------------------------------------------------------------------------
__device__ uint4 tile_carry_meta(const Geometry &g, const int32_t *__restrict__ row_ptr,
                                 const uint32_t *__restrict__ tile_ptr, const int t,
                                 uint32_t *__restrict__ long_run_counter)
{
    const long long T = g.tile_elems;
    // is tile k the head of a short-spill run?  (needs k >= 1)
    auto short_spill = [&](int k, int *len) -> bool {
        if (k < 1 || k >= g.p)
            return false;
        const int r = (int)(tile_ptr[k] & ROW_MASK);
        if ((int)(tile_ptr[k - 1] & ROW_MASK) == r)
            return false; // not a run head
        if ((long long)row_ptr[r] == (long long)k * T)
            return false; // row starts on the boundary: nothing spills
        if (k + 1 < g.p && (int)(tile_ptr[k + 1] & ROW_MASK) == r)
            return false; // row runs on into tile k+1
        const long long L = (long long)row_ptr[r + 1] - (long long)k * T;
        if (L > OMEGA)
            return false;
        *len = (int)L;
        return true;
    };
    const int r = (int)(tile_ptr[t] & ROW_MASK);
    // Run bounds on the sorted tile_ptr: head = first tile whose row is r, e = last one.  Galloping from t (steps 1, 2,
    // 4, ...) and a bisection of the last stride: one dependent load for the usual run of one tile, log(run) for a hub
    // row (a thread used to WALK its run: ten thousand serial steps for a 10 M-nnz row; a plain bisection of [0, t] is
    // twelve dependent loads on EVERY tile of a 2 500-tile matrix, 6 us of a 60-us conversion).
    int hi = t, lo = t - 1;
    for (int step = 1; lo >= 0 && (int)(tile_ptr[lo] & ROW_MASK) >= r; step <<= 1) {
        hi = lo;
        lo -= step;
    }
    lo = lo < 0 ? 0 : lo + 1; // every tile below lo has a smaller row; tile hi has row r
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if ((int)(tile_ptr[mid] & ROW_MASK) < r)
            lo = mid + 1;
        else
            hi = mid;
    }
    const int head_idx = lo;
    const bool head = head_idx == t;
    uint4 meta = make_uint4(0u, (unsigned)head_idx, 0u, 0u);
    int len = 0;
    if (!g.defer && short_spill(t, &len)) {
        meta.x |= 1u << 28;
    } else if (head && r < g.m) {
        lo = t + 1, hi = t + 1; // first tile after t whose row is larger (p if none), galloping forward
        for (int step = 1; hi < g.p && (int)(tile_ptr[hi] & ROW_MASK) <= r; step <<= 1) {
            lo = hi + 1;
            hi += step;
        }
        hi = hi > g.p ? g.p : hi;
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if ((int)(tile_ptr[mid] & ROW_MASK) <= r)
                lo = mid + 1;
            else
                hi = mid;
        }
        const int e = lo - 1;
        unsigned expected = (unsigned)(e - t + 1);
        if (e - t + 1 > RUN_SERIAL_MAX) {
            meta.x |= 1u << 26; // long run: partials are only parked, k_calibrate sums them
            if (long_run_counter)
                atomicAdd(long_run_counter, 1u);
        }
        if ((long long)row_ptr[r] != (long long)t * T) {
            expected += 1; // row r starts inside tile t-1, whose closing segment also arrives
            meta.x |= 1u << 27;
        }
        if (g.defer && expected >= 2u)
            meta.x |= 1u << 26; // deferred carries: every meeting of partials is finished by k_calibrate
        meta.x |= expected;
    }
    if (t + 1 < g.p) {
        const int rn = (int)(tile_ptr[t + 1] & ROW_MASK);
        // the closing segment of tile t belongs to row rn iff rn starts before (t+1)*T
        if (rn != r && (long long)row_ptr[rn] != (long long)(t + 1) * T) {
            meta.x |= 1u << 30;
            if (!g.defer && short_spill(t + 1, &len)) {
                meta.x |= 1u << 29;
                meta.z = (unsigned)len;
            }
        }
    }
    return meta; // .w (x-window start) is filled in by k_tile_tables
}

// ---------------------------------------------------------------------------------------------
// x-window selection (ours, not part of the reference format).  One wavefront per tile t < p-1.
// Candidates = the 64 column indices of the tile's first step (one per lane, a uniform sample of the
// tile).  Each candidate centres a window of XWIN_BYTES / sizeof(vT) columns and is scored by how many
// of the 64 samples fall inside it (64 readlane broadcasts); the best-scoring candidate is then
// verified against ALL omega*sigma elements.  If at least XWIN_MIN_COVER_PCT % of them fall inside,
// carry_meta[t].w = window start + 1 (0 = no window); counters[0] counts such tiles, counters[1] the
// non-zeros they cover.  The SpMV kernel stages that slice of x in LDS and serves the
// in-window gathers from LDS (csr5_spmv.hip, XWIN variant).  Two passes over the tile's columns.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int wave_sum_i32(int v)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        v += __shfl_xor(v, d, OMEGA);
    return v;
}

// One wavefront per tile t < p writes everything the SpMV kernels keep per tile besides the format arrays:
// carry_meta[t] (lane 0 derives it, lane 1 derives tile t+1's first word at the same time: the two bisection chains
// run side by side under the tile's column loads), the x-window of the tile, and the 32-byte header of the fused kernel
// -- words 0..3 = carry_meta[t], 4 = carry_meta[t+1].x, 5..6 = tile_ptr[t], tile_ptr[t+1] (copies; the format arrays
// themselves stay untouched), 7 = the tile's window statistics (bits 0..15 covered non-zeros, 16..31 lines), which
// only k_stats_export reads.  (Three launches -- carry_meta, tile_window, tile_hdr -- until round 2.)
__global__ void __launch_bounds__(FMT_BLOCK) k_tile_tables(Geometry g, const int32_t *__restrict__ row_ptr,
                                                       const uint32_t *__restrict__ tile_ptr,
                                                       const int32_t *__restrict__ col,
                                                       uint4 *__restrict__ carry_meta, uint32_t *__restrict__ hdr,
                                                       uint32_t *__restrict__ counters, int XWIN_ELEMS,
                                                       int line_shift)
{
    stamp_phase(counters, 3);
    uint32_t *const long_run_counter = counters + 2;
    const int lane = threadIdx.x & (OMEGA - 1);
    const int t = blockIdx.x * FMT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (t >= g.p)
        return;
    const bool windowed = t < g.p - 1; // the CSR tail has no window
    const int32_t *c = col + (size_t)t * g.tile_elems + lane;
    const int sample = windowed ? c[0] : 0;

    uint4 meta = make_uint4(0u, 0u, 0u, 0u);
    if (lane < 2 && t + lane < g.p)
        meta = tile_carry_meta(g, row_ptr, tile_ptr, t + lane, lane == 0 ? long_run_counter : nullptr);
    const unsigned next_x = (unsigned)__builtin_amdgcn_readlane((int)meta.x, 1);

    // window selection for XE columns: {first column, non-zeros inside} of the best candidate, {-1, 0} if it covers less than
    // XWIN_MIN_COVER_PCT % of the tile
    auto pick_window = [&](int XE, int *inside_out) -> int {
        const int hi_limit = g.n > XE ? g.n - XE : 0;
        auto window_of = [&](int centre) {
            int lo = centre - XE / 2;
            lo = lo < 0 ? 0 : (lo > hi_limit ? hi_limit : lo);
            return lo & ~3;
        };
        // score every lane's candidate on the 64 samples (v_readlane broadcasts, no LDS shuffles)
        const int my_lo = window_of(sample);
        int score = 0;
#pragma unroll
        for (int j = 0; j < OMEGA; j++)
            score += (unsigned)(__builtin_amdgcn_readlane(sample, j) - my_lo) < (unsigned)XE;
        // best candidate: highest score, lowest lane on ties (deterministic)
        int best = score * OMEGA + (OMEGA - 1 - lane);
#pragma unroll
        for (int d = 32; d >= 1; d >>= 1) {
            const int o = __shfl_xor(best, d, OMEGA);
            best = o > best ? o : best;
        }
        const int best_all = __builtin_amdgcn_readfirstlane(best);
        const int lo = __builtin_amdgcn_readlane(my_lo, OMEGA - 1 - (best_all % OMEGA));
        const int best_score = best_all / OMEGA;
        *inside_out = 0;
        // The 64 samples are every sigma-th element of the tile.  A window that holds fewer than a quarter of the samples
        // it would need cannot cover XWIN_MIN_COVER_PCT % of the tile: such tiles (every tile of a matrix with scattered
        // columns) are left without reading their other column words -- 1 GB and 0.3 ms of this pass on R-MAT 24.
        if (best_score * 400 < OMEGA * XWIN_MIN_COVER_PCT)
            return -1;
        int inside = 0;
        for (int i = 0; i < g.sigma; i++)
            inside += (unsigned)(c[i * OMEGA] - lo) < (unsigned)XE;
        inside = wave_sum_i32(inside);
        if (inside * 100 < g.tile_elems * XWIN_MIN_COVER_PCT)
            return -1;
        *inside_out = inside;
        return lo;
    };
    unsigned window = 0, stats = 0;
    if (windowed) {
        int inside = 0;
        const int lo = pick_window(XWIN_ELEMS, &inside);
        if (lo >= 0) {
            // How many distinct 128-byte lines of x do the in-window lanes of ONE gather instruction (the 64 samples)
            // touch?  That is what the window replaces: a gather spread over 30+ lines costs 60+ clk in the
            // vector-memory path, one that sits on a handful of lines is cheap and the staging would cost more than
            // it saves.
            const bool in_win = (unsigned)(sample - lo) < (unsigned)XWIN_ELEMS;
            const int line = sample >> line_shift;
            bool first = in_win;
#pragma unroll
            for (int j = 0; j < OMEGA - 1; j++) {
                const int other = __builtin_amdgcn_readlane(line, j);
                const bool other_in = (__builtin_amdgcn_readlane((int)in_win, j) != 0);
                first = first && !(j < lane && other_in && other == line);
            }
            const int lines = __popcll(__ballot(first));
            window = (unsigned)lo + 1u;
            stats = (unsigned)inside | ((unsigned)lines << 16);
        }
    }
    if (lane == 0) {
        meta.w = window;
        carry_meta[t] = meta;
        reinterpret_cast<uint4 *>(hdr)[2 * (size_t)t] = meta;
        reinterpret_cast<uint4 *>(hdr)[2 * (size_t)t + 1] = make_uint4(next_x, tile_ptr[t], tile_ptr[t + 1], stats);
    }
}

// LAST kernel of the conversion.  Sums the per-tile window statistics (counters[0] = tiles with a window, [1] =
// non-zeros inside those windows, [3] = distinct x lines under the in-window lanes of the sampled gathers; per-tile
// words because two global atomics per tile on the same two words serialised a whole kernel: 646 us for 28 k tiles),
// and the workgroup that finishes last exports the six words the host needs -- tail start, number of offsets
// (anonymouslib_cuda.h:165-167, format_cuda.h:331-343) and the four statistics -- straight into pinned,
// device-visible host memory: everything it reads is final, and the host reads them after its one synchronisation
// without three 4-to-16-byte device-to-host copies (7.6 us each).
__global__ void __launch_bounds__(1024) k_stats_export(Geometry g, const uint32_t *__restrict__ tile_ptr,
                                                       const uint32_t *__restrict__ hdr,
                                                       const int32_t *__restrict__ offset_ptr,
                                                       uint32_t *__restrict__ counters,
                                                       uint32_t *__restrict__ host_words, int export_only)
{
    __shared__ unsigned part[16][3];
    unsigned on = 0, in = 0, lines = 0;
    if (export_only)
        stamp_phase(counters, 3); // (k_tile_tables, which stamps this phase, did not run)
    for (int t = blockIdx.x * 1024 + threadIdx.x; t < (export_only ? 0 : g.p - 1); t += gridDim.x * 1024) {
        const unsigned v = hdr[8 * (size_t)t + 7];
        on += v != 0;
        in += v & 0xFFFFu;
        lines += v >> 16;
    }
    on = (unsigned)wave_sum_i32((int)on);
    in = (unsigned)wave_sum_i32((int)in);
    lines = (unsigned)wave_sum_i32((int)lines);
    if ((threadIdx.x & (OMEGA - 1)) == 0) {
        part[threadIdx.x >> 6][0] = on;
        part[threadIdx.x >> 6][1] = in;
        part[threadIdx.x >> 6][2] = lines;
    }
    __syncthreads();
    if (threadIdx.x != 0)
        return;
    on = in = lines = 0;
    for (int w = 0; w < 16; w++)
        on += part[w][0], in += part[w][1], lines += part[w][2];
    if (gridDim.x > 1) { // (one workgroup up to 16 k tiles: no atomics, no fence)
        if (on | in) {
            atomicAdd(counters + 0, on);
            atomicAdd(counters + 1, in);
            atomicAdd(counters + 3, lines);
        }
        __threadfence();
        if (atomicAdd(counters + 4, 1u) + 1u != gridDim.x)
            return; // not the last workgroup
        counters[4] = 0;
        on = __hip_atomic_load(counters + 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        in = __hip_atomic_load(counters + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lines = __hip_atomic_load(counters + 3, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (!host_words)
        return;
    uint4 *out = reinterpret_cast<uint4 *>(host_words);
    const uint4 *stamps = reinterpret_cast<const uint4 *>(counters + STAMP_WORD);
    out[0] = make_uint4(tile_ptr[g.p - 1], (uint32_t)offset_ptr[g.p], on, in);
    out[1] = make_uint4(counters[2], lines, 0u, 0u);
    out[2] = stamps[0]; // phase stamps 0, 1 (64-bit each)
    out[3] = stamps[1]; // phase stamps 2, 3
}

// ---------------------------------------------------------------------------------------------
// Narrow column codes (ours; a kernel-side table like the x-window, the format arrays are untouched).  A tile of a banded /
// blocked matrix spans far fewer than 2^15 columns: column - (smallest column of the tile) fits 15 bits, the 16th carries the
// element's row-start flag, and the x-window kernel then streams 2 bytes per non-zero instead of 4 and no descriptor words
// (fp32: 6 instead of 8.25 bytes per non-zero in all).  One wavefront per
// tile t < p-1: minimum and maximum of the tile's column words, the codes of a lane's elements (2 dd, 2 dd + 1) in its word dd
// (element i of lane l = position i * 64 + l of the tile-ordered column_index, so a code pairs with the value there whether or
// not the tile was transposed), a lane's words in 16-byte pieces, base16[t]; a tile that spans 32 768 columns or more counts
// into *wide_tiles -- the codes are used only when that stays 0.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(FMT_BLOCK) k_col16(Geometry g, const int32_t *__restrict__ col,
                                                 const uint32_t *__restrict__ tile_desc, uint32_t *__restrict__ col16,
                                                 int32_t *__restrict__ base16, uint32_t *__restrict__ wide_tiles)
{
    const int lane = threadIdx.x & (OMEGA - 1);
    const int t = blockIdx.x * FMT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (t >= g.p - 1)
        return;
    const int32_t *c = col + (size_t)t * g.tile_elems + lane;
    int lo = 0x7FFFFFFF, hi = 0;
    for (int i = 0; i < g.sigma; i++) {
        const int v = c[i * OMEGA];
        lo = v < lo ? v : lo;
        hi = v > hi ? v : hi;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int ol = __shfl_xor(lo, d, OMEGA), oh = __shfl_xor(hi, d, OMEGA);
        lo = ol < lo ? ol : lo;
        hi = oh > hi ? oh : hi;
    }
    if (lane == 0) {
        base16[t] = lo;
        if (hi - lo >= COL16_SPAN)
            atomicAdd(wide_tiles, 1u);
    }
    // word dd of a lane (codes of its elements 2 dd, 2 dd + 1): the lane's words in 16-byte pieces, piece k of all lanes
    // together (the kernel reads a piece with ONE 16-byte load per lane, 1 KB per wave instruction); sigma / 2 not a multiple
    // of 4 (sigma = 12): the last two words as an 8-byte piece
    // bit 15 of a code = the element starts a row: the lane's bit flags of the reference's descriptor (format_cuda.h:129-360,
    // element i -> bit 31 - i once the y_offset / scansum fields are shifted out), so that the kernel reads no descriptor word
    const uint32_t *dw = tile_desc + (size_t)t * OMEGA * g.num_packet + lane;
    uint32_t flags = dw[0] << g.bit_all;
    if (g.num_packet > 1)
        flags |= dw[OMEGA] >> (32 - g.bit_all);
    uint32_t *out = col16 + (size_t)t * (g.tile_elems / 2);
    const int W = g.sigma / 2, G4 = W / 4;
    for (int dd = 0; dd < W; dd++) {
        const uint32_t a = (uint32_t)(c[(2 * dd) * OMEGA] - lo), b = (uint32_t)(c[(2 * dd + 1) * OMEGA] - lo);
        const uint32_t fa = (flags >> (31 - 2 * dd)) & 1u, fb = (flags >> (30 - 2 * dd)) & 1u;
        const int at = dd < 4 * G4 ? (dd >> 2) * 4 * OMEGA + lane * 4 + (dd & 3) : G4 * 4 * OMEGA + lane * 2 + (dd - 4 * G4);
        out[at] = (a & 0x7FFFu) | (fa << 15) | ((b & 0x7FFFu) << 16) | (fb << 31);
    }
}

hipError_t launch_col16(const Geometry &g, const DeviceArrays &d, uint32_t *col16, int32_t *base16, uint32_t *wide_tiles,
                        hipStream_t s)
{
    if (g.p <= 1)
        return hipSuccess;
    hipLaunchKernelGGL(k_col16, dim3((g.p - 1 + FMT_WAVES_PER_BLOCK - 1) / FMT_WAVES_PER_BLOCK), dim3(FMT_BLOCK), 0, s, g, d.col, d.tile_desc, col16, base16,
                       wide_tiles);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// Flagged column words (csr5_spmv.hip C31): the tile-ordered column_index of tiles 0 .. p-2 with the element's row-start flag --
// the lane's bit flags of the reference's descriptor (format_cuda.h:129-360) -- in bit 31.  The caller's column_index (one of
// the reference's format arrays, exposed bit for bit) is only read.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(FMT_BLOCK) k_col31(Geometry g, const int32_t *__restrict__ col,
                                                 const uint32_t *__restrict__ tile_desc, uint32_t *__restrict__ col31)
{
    const int lane = threadIdx.x & (OMEGA - 1);
    const int t = blockIdx.x * FMT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (t >= g.p - 1)
        return;
    const uint32_t *dw = tile_desc + (size_t)t * OMEGA * g.num_packet + lane;
    uint32_t flags = dw[0] << g.bit_all;
    if (g.num_packet > 1)
        flags |= dw[OMEGA] >> (32 - g.bit_all);
    const size_t base = (size_t)t * g.tile_elems + lane;
    for (int i = 0; i < g.sigma; i++)
        col31[base + (size_t)i * OMEGA] = (uint32_t)col[base + (size_t)i * OMEGA] | (((flags >> (31 - i)) & 1u) << 31);
}

hipError_t launch_col31(const Geometry &g, const DeviceArrays &d, uint32_t *col31, hipStream_t s)
{
    if (g.p <= 1)
        return hipSuccess;
    hipLaunchKernelGGL(k_col31, dim3((g.p - 1 + FMT_WAVES_PER_BLOCK - 1) / FMT_WAVES_PER_BLOCK), dim3(FMT_BLOCK), 0, s, g, d.col, d.tile_desc, col31);
    return hipGetLastError();
}

// checkpoint loading: the index arrays come from a file and are used as addresses by every later kernel
__global__ void __launch_bounds__(256) k_validate_csr(int m, int n, int nnz, const int32_t *__restrict__ row_ptr,
                                                      const int32_t *__restrict__ col, uint32_t *__restrict__ flag)
{
    uint32_t bad = 0;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i <= (size_t)m; i += stride) {
        const int a = row_ptr[i];
        if (i == 0 && a != 0)
            bad |= 1u;
        if (i == (size_t)m ? a != nnz : (a > row_ptr[i + 1] || a < 0))
            bad |= 1u;
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < (size_t)nnz; i += stride)
        if ((uint32_t)col[i] >= (uint32_t)n)
            bad |= 2u;
    if (bad)
        atomicOr(flag, bad);
}

__global__ void k_warmup(int *out)
{
    __shared__ int s[OMEGA];
    s[threadIdx.x] = threadIdx.x;
    __syncthreads();
    int v = 0;
    for (int i = 0; i < 50; i++)
        v += s[(threadIdx.x + i) & (OMEGA - 1)];
    if (v == -1)
        out[0] = v;
}

// ---------------------------------------------------------------------------------------------
// host launchers
// ---------------------------------------------------------------------------------------------
static inline int div_up(long long a, int b) { return (int)((a + b - 1) / b); }

hipError_t launch_row_scan(const Geometry &g, const DeviceArrays &d, hipStream_t s)
{
    if (g.m <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_row_scan, dim3(div_up(g.m, FMT_BLOCK)), dim3(FMT_BLOCK), 0, s, g, d.row_ptr,
                       d.tile_ptr, d.tile_desc, d.counters);
    return hipGetLastError();
}

hipError_t launch_tile_desc(const Geometry &g, const DeviceArrays &d, hipStream_t s)
{
    if (g.p <= 1)
        return hipSuccess;
    hipLaunchKernelGGL(k_tile_desc, dim3(div_up(g.p - 1, FMT_WAVES_PER_BLOCK)), dim3(FMT_BLOCK), 0, s, g, d.row_ptr,
                       d.tile_ptr, d.tile_desc, d.offset_ptr, d.counters);
    return hipGetLastError();
}

// K6 for small tile counts: the exclusive scan of offset_ptr[0..entries) by ONE workgroup in ONE launch (thread i owns
// `per` consecutive entries; wave shuffles + 16 LDS words carry the rest).  rocprim's device scan is two launches
// (look-back state initialisation + scan) and takes over above SMALL_SCAN_MAX entries, where it is the faster one.
constexpr int SMALL_SCAN_MAX = 16384;
__global__ void __launch_bounds__(1024) k_offset_scan_small(int32_t *__restrict__ a, int entries)
{
    __shared__ int wave_total[16];
    const int per = (entries + 1023) / 1024;
    const int first = (int)threadIdx.x * per;
    int sum = 0;
    for (int i = first; i < first + per && i < entries; i++)
        sum += a[i];
    const int lane = threadIdx.x & (OMEGA - 1), w = threadIdx.x >> 6;
    int incl = sum;
#pragma unroll
    for (int d = 1; d < OMEGA; d <<= 1) {
        const int v = __shfl_up(incl, d, OMEGA);
        if (lane >= d)
            incl += v;
    }
    if (lane == OMEGA - 1)
        wave_total[w] = incl;
    __syncthreads();
    int run = incl - sum;
    for (int k = 0; k < w; k++)
        run += wave_total[k];
    for (int i = first; i < first + per && i < entries; i++) {
        const int v = a[i];
        a[i] = run;
        run += v;
    }
}

size_t offset_scan_tmp_bytes(int entries)
{
    size_t bytes = 0;
    int32_t *null_i = nullptr;
    (void)rocprim::exclusive_scan(nullptr, bytes, null_i, null_i, 0, (size_t)entries, rocprim::plus<int32_t>(), nullptr);
    return bytes;
}

hipError_t launch_offset_scan(const Geometry &g, const DeviceArrays &d, void *tmp, size_t tmp_bytes, hipStream_t s)
{
    if (g.p + 1 <= SMALL_SCAN_MAX) {
        hipLaunchKernelGGL(k_offset_scan_small, dim3(1), dim3(1024), 0, s, d.offset_ptr, g.p + 1);
        return hipGetLastError();
    }
    return rocprim::exclusive_scan(tmp, tmp_bytes, d.offset_ptr, d.offset_ptr, 0, (size_t)g.p + 1, rocprim::plus<int32_t>(), s);
}

hipError_t launch_desc_offset(const Geometry &g, const DeviceArrays &d, hipStream_t s)
{
    if (g.p <= 1)
        return hipSuccess;
    hipLaunchKernelGGL(k_desc_offset, dim3(div_up(g.p - 1, FMT_WAVES_PER_BLOCK)), dim3(FMT_BLOCK), 0, s, g,
                       d.row_ptr, d.tile_ptr, d.tile_desc, d.offset_ptr, d.offset);
    return hipGetLastError();
}

// Values of a packed hot child (every tile 0 .. p-2, values only) without LDS: lane l of a wavefront reads its SIGMA
// consecutive values (CSR order, 16-byte pieces) and the wavefront stores them step by step, coalesced.  One wavefront per
// tile, no synchronisation; the LDS version above moves the same 2 x 2.1 GB of R-MAT 24 in 0.97 ms.
template <typename VT, int SIGMA>
__global__ void __launch_bounds__(FMT_BLOCK) k_transpose_values(Geometry g, VT *__restrict__ val, uint32_t *__restrict__ counters)
{
    stamp_phase(counters, 2);
    constexpr int T = OMEGA * SIGMA;
    const int lane = threadIdx.x & (OMEGA - 1);
    const int t = blockIdx.x * FMT_WAVES_PER_BLOCK + (threadIdx.x >> 6);
    if (t >= g.p - 1)
        return;
    VT *tile = val + (size_t)t * T;
    VT v[SIGMA];
    constexpr int PER = 16 / sizeof(VT); // values per 16-byte piece
    typedef VT piece_t __attribute__((ext_vector_type(PER)));
    static_assert(SIGMA % PER == 0, "whole 16-byte pieces per lane");
#pragma unroll
    for (int q = 0; q < SIGMA / PER; q++) {
        const piece_t w = *reinterpret_cast<const piece_t *>(tile + lane * SIGMA + q * PER);
#pragma unroll
        for (int e = 0; e < PER; e++)
            v[q * PER + e] = w[e];
    }
    // every load of the tile has returned before its first store goes out (the tile is private to this wavefront, but
    // lane A's store lands where lane B reads)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // lane-major 16-BYTE PIECES (not the reference's element-wise transposition: this array is internal to the slab child and
    // only k_spmv_range reads it): piece q of lane l -- its elements q PER .. q PER + PER - 1 -- at (q * 64 + l) * PER, so the
    // range kernel fetches a lane's sigma values with sigma / PER 16-byte loads instead of sigma narrow ones (same lines, half
    // / a quarter of the vector-memory instructions: -1.8 % in the shape probe, profiles/r04_probes.txt)
#pragma unroll
    for (int q = 0; q < SIGMA / PER; q++) {
        piece_t w;
#pragma unroll
        for (int e = 0; e < PER; e++)
            w[e] = v[q * PER + e];
        *reinterpret_cast<piece_t *>(tile + ((size_t)q * OMEGA + lane) * PER) = w;
    }
}

hipError_t launch_transpose(const Geometry &g, const DeviceArrays &d, int value_type, bool r2c, hipStream_t s)
{
    if (g.p <= 1)
        return hipSuccess;
    const size_t vsz = value_type == CSR5HIP_F64 ? 8 : 4;
    // ~12 KB of tile data per workgroup (one tile at sigma = 16 fp64, two at sigma = 8, ...)
    int tpb = (int)(12288 / ((size_t)g.tile_elems * (vsz + 4)));
    tpb = tpb < 1 ? 1 : (tpb > 16 ? 16 : tpb);
    const size_t lds = (size_t)tpb * g.sigma * transpose_pitch(g.sigma) * (vsz + 4);
    const dim3 grid((unsigned)((g.p - 1 + tpb - 1) / tpb));
    if (value_type == CSR5HIP_F64)
        hipLaunchKernelGGL(k_transpose<double>, grid, dim3(FMT_BLOCK), lds, s, g, d.tile_ptr, d.col, (double *)d.val, r2c ? 1 : 0, tpb,
                           r2c ? d.counters : nullptr);
    else
        hipLaunchKernelGGL(k_transpose<float>, grid, dim3(FMT_BLOCK), lds, s, g, d.tile_ptr, d.col, (float *)d.val, r2c ? 1 : 0, tpb,
                           r2c ? d.counters : nullptr);
    return hipGetLastError();
}

hipError_t launch_transpose_values(const Geometry &g, const DeviceArrays &d, int value_type, hipStream_t s)
{
    if (g.p <= 1)
        return hipSuccess;
    const dim3 grid((unsigned)div_up(g.p - 1, FMT_WAVES_PER_BLOCK)), block(FMT_BLOCK);
#define CSR5_TV(VT, S)                                                                                                \
    hipLaunchKernelGGL((k_transpose_values<VT, S>), grid, block, 0, s, g, (VT *)d.val, d.counters);                  \
    return hipGetLastError();
    if (value_type == CSR5HIP_F64) {
        switch (g.sigma) {
        case 8: CSR5_TV(double, 8)
        case 16: CSR5_TV(double, 16)
        }
    } else {
        switch (g.sigma) {
        case 8: CSR5_TV(float, 8)
        case 16: CSR5_TV(float, 16)
        }
    }
#undef CSR5_TV
    return hipErrorInvalidValue; // (a hot child's sigma is 8 or 16: csr5_internal.h hot_child_sigma)
}

// export_only: the matrix is a column-slab child served by the range kernel (csr5_hot.hip), which uses neither carry meta nor
// tile headers nor x-windows: only the host's words are exported (k_tile_tables reads every column word: 0.86 ms of the
// 3.2-ms child conversion of R-MAT 24)
hipError_t launch_tile_tables(const Geometry &g, const DeviceArrays &d, int value_size, uint32_t *host_words, bool export_only,
                              hipStream_t s)
{
    if (g.p <= 0)
        return hipSuccess;
    if (export_only) {
        hipLaunchKernelGGL(k_stats_export, dim3(1), dim3(1024), 0, s, g, d.tile_ptr, d.tile_hdr, d.offset_ptr, d.counters,
                           host_words, 1);
        return hipGetLastError();
    }
    hipLaunchKernelGGL(k_tile_tables, dim3(div_up(g.p, FMT_WAVES_PER_BLOCK)), dim3(FMT_BLOCK), 0, s, g, d.row_ptr,
                       d.tile_ptr, d.col, reinterpret_cast<uint4 *>(d.carry_meta), d.tile_hdr, d.counters,
                       xwin_elems(value_size), value_size == 8 ? 4 : 5);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    int blocks = div_up(g.p, 1024 * 16);
    blocks = blocks < 1 ? 1 : (blocks > 64 ? 64 : blocks);
    hipLaunchKernelGGL(k_stats_export, dim3(blocks), dim3(1024), 0, s, g, d.tile_ptr, d.tile_hdr, d.offset_ptr,
                       d.counters, host_words, 0);
    return hipGetLastError();
}

hipError_t launch_validate_csr(int m, int n, int nnz, const int32_t *row_ptr, const int32_t *col, uint32_t *flag,
                               hipStream_t s)
{
    long long blocks = ((long long)(nnz > m ? nnz : m + 1) + 256 * 8 - 1) / (256 * 8);
    blocks = blocks < 1 ? 1 : (blocks > 16384 ? 16384 : blocks);
    hipLaunchKernelGGL(k_validate_csr, dim3((unsigned)blocks), dim3(256), 0, s, m, n, nnz, row_ptr, col, flag);
    return hipGetLastError();
}

hipError_t launch_warmup(hipStream_t s)
{
    hipLaunchKernelGGL(k_warmup, dim3(4000), dim3(OMEGA), 0, s, (int *)nullptr);
    return hipGetLastError();
}

} // namespace csr5

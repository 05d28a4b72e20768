// csr5_slab.hip -- column slabs: a kernel-side acceleration structure for matrices whose x does not fit one
// XCD's 4-MB L2 and whose columns are scattered (power-law graphs).  Ours; not part of the reference format: the
// reference streams x through one texture/L2 (CSR5_cuda/detail/cuda/csr5_spmv_cuda.h:7-23 `candidate`), and the
// four CSR5 arrays the handle exposes stay exactly what the reference's format code produces.
//
// Why: a random 8-byte x gather that misses L2 pulls a whole 128-byte line through the fabric; with eight XCDs
// each caching the SAME hot part of x in its own 4-MB L2, R-MAT / web-graph inputs moved 3.4-3.9x their
// algorithmic bytes (profiles/r01_*).  Here the columns are hashed (xor-fold of col >> shift) into S slabs and the
// non-zeros are stably partitioned by slab; inside a slab they keep their CSR order, so every (row, slab) pair with
// at least one non-zero is a contiguous SEGMENT.  Stacking the S sub-matrices vertically gives ONE CSR matrix A'
// with m' = number of segments rows, the same n and the same non-zeros -- which is converted to CSR5 and multiplied
// by the ordinary tile kernel (an internal child handle).  With XCD-contiguous tile ranges every XCD works on its
// own slab(s): the eight L2s hold eight DIFFERENT parts of x.  P = A' x holds one partial per segment, slab after slab
// and row after row inside a slab, so the partials of slab k that belong to the 256 rows of a row block are ONE
// contiguous run of P.  k_slab_combine gives every wavefront a row block: it streams the block's S runs (run bounds:
// `base`, S words per block; the row of every partial inside its block: `rowidx`, one byte per segment) and adds them
// into 256 accumulators in LDS, slab after slab.  Deterministic: partials are added in slab order, no floating-point
// atomics; rows without non-zeros are not written (`nonempty`, one bit per row).
//
// Build (all on the device, reading the parent's tile-ordered column_index / value through the transpose map):
//   k_slab_hist     workgroup / tile: slab histogram of the tile                       -> hist[slab][tile]
//   exclusive scan  (rocprim) over hist in slab-major order = start of every (slab, tile) chunk in A'
//   k_slab_scatter  workgroup / tile: stable rank of every element inside its slab (wave ballots + LDS chunk table),
//                   scatter of column, value and the 32-bit key (row, bit 31 = first element of a slab)
//   segment starts  = positions where the key changes: counted per chunk, scanned, then written into row_ptr' by
//                   ballot rank (k_slab_count_segments, k_slab_scan_counts, k_slab_emit_segments)
//                   (the emit pass also writes the row of every segment inside its 256-row block, one byte)
//   k_slab_base     thread / (row block, slab): first segment of the run (binary search on the sorted segment keys)
//   k_slab_nonempty thread / row: one bit per row of the parent
#include "csr5_internal.h"
#include "csr5_slabmap.h"

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/device/device_scan.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace csr5 {

constexpr int SLAB_BLOCK = 256;
constexpr int SLAB_MAX = 64;
constexpr uint32_t SLAB_KEY_FIRST = 0x80000000u; // key bit: first element of a slab

// 128-bit groups of one slab's bitmap
__host__ __device__ inline size_t slab_hot_groups(int n, int shift, int bits) { return (slab_local_count(n, shift, bits) + 127) / 128; }
// set bits of a group below position `bit`, and whether `bit` itself is set
__device__ __forceinline__ bool hot_lookup(const uint4 w, uint32_t bit, uint32_t &below)
{
    const unsigned long long lo = (unsigned long long)w.x | ((unsigned long long)w.y << 32);
    const unsigned long long hi = (unsigned long long)w.z | ((unsigned long long)w.w << 32);
    const unsigned long long word = bit < 64 ? lo : hi;
    const uint32_t b = bit & 63;
    below = (uint32_t)__popcll(word & ((1ull << b) - 1ull)) + (bit < 64 ? 0u : (uint32_t)__popcll(lo));
    return (word >> b) & 1ull;
}

__device__ __forceinline__ int upper_bound_i32(const int32_t *__restrict__ a, int key, int size)
{
    int lo = 0, hi = size;
    while (lo < hi) {
        const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if (a[mid] <= key)
            lo = mid + 1;
        else
            hi = mid;
    }
    return lo;
}

// tile t of the parent: storage position q <-> CSR rank c (format_cuda.h:525-744: element (lane l, step i) of a
// transposed tile sits at i*omega + l and is the (l*sigma + i)-th element of the tile in CSR order)
__device__ __forceinline__ bool tile_is_transposed(const Geometry &g, const uint32_t *tile_ptr, int t)
{
    return t < g.p - 1 && tile_ptr[t] != tile_ptr[t + 1];
}

// One wavefront per tile, HIST_TILES consecutive tiles per workgroup.  Every LANE counts its own elements in LDS counters
// of its own ([slab][lane], padded to 65 lanes so that the column sums below do not collide on one bank): one ds_add per
// element and no cross-lane work; lane k then adds up row k.  (64 LDS atomics on 16 shared words per chunk serialised;
// counting by ballots -- one per bit of the slab id, intersected per lane -- was bound by its ~45 vector instructions per
// chunk.)  A workgroup writes HIST_TILES consecutive words per slab (16 strided words per tile were 4 M partial lines).
constexpr int HIST_TILES = 16, HIST_PAD = OMEGA + 1;
__global__ void __launch_bounds__(SLAB_BLOCK)
k_slab_hist(Geometry g, const int32_t *__restrict__ col, int S, int bits, int shift, uint32_t *__restrict__ hist)
{
    extern __shared__ uint32_t hist_lds[];
    __shared__ uint32_t sh[SLAB_MAX][HIST_TILES];
    const int lane = threadIdx.x & (OMEGA - 1), wave = threadIdx.x >> 6;
    uint32_t *mine = hist_lds + (size_t)wave * S * HIST_PAD;
    const int t0 = blockIdx.x * HIST_TILES;
    for (int i = wave; i < HIST_TILES; i += SLAB_BLOCK / OMEGA) {
        const int t = t0 + i;
        uint32_t total = 0;
        if (t < g.p) {
            for (int k = 0; k < S; k++)
                mine[k * HIST_PAD + lane] = 0;
            const size_t base = (size_t)t * g.tile_elems;
            const int E = (int)((size_t)g.nnz - base < (size_t)g.tile_elems ? (size_t)g.nnz - base : (size_t)g.tile_elems);
            constexpr int AHEAD = 8; // chunks requested together
            for (int q0 = 0; q0 < E; q0 += AHEAD * OMEGA) {
                int32_t c[AHEAD];
#pragma unroll
                for (int u = 0; u < AHEAD; u++) {
                    const int q = q0 + u * OMEGA + lane;
                    c[u] = col[base + (q < E ? q : E - 1)];
                }
#pragma unroll
                for (int u = 0; u < AHEAD; u++)
                    if (q0 + u * OMEGA + lane < E)
                        atomicAdd(&mine[slab_of((uint32_t)c[u], shift, bits) * HIST_PAD + lane], 1u);
            }
            // (one wavefront: its LDS operations are executed in order)
            if (lane < S)
                for (int j = 0; j < OMEGA; j++)
                    total += mine[lane * HIST_PAD + j];
        }
        if (lane < S)
            sh[lane][i] = total;
    }
    __syncthreads();
    for (int e = threadIdx.x; e < S * HIST_TILES; e += SLAB_BLOCK) {
        const int k = e / HIST_TILES, i = e % HIST_TILES;
        if (t0 + i < g.p)
            hist[(size_t)k * g.p + t0 + i] = sh[k][i];
    }
}

// lanes of the wavefront that hold the same slab as this lane (valid lanes only): one ballot per bit of the slab id
__device__ __forceinline__ unsigned long long same_slab_lanes(int k, bool valid, int bits)
{
    unsigned long long same = __ballot(valid);
    for (int b = 0; b < bits; b++) {
        const bool on = (k >> b) & 1;
        const unsigned long long bal = __ballot(on);
        same &= on ? bal : ~bal;
    }
    return same;
}

// One workgroup per tile of the parent.  The tile is read once (coalesced, through the transpose map) into LDS in CSR
// order, ranked per slab (stable), and written out SLAB AFTER SLAB: consecutive lanes write consecutive elements of one
// slab's run, so a tile produces S contiguous runs per array.  (Writing straight from CSR order -- every 64-element chunk
// scattering ~4 elements to each of 16 slabs -- issued 16 partial-line writes per instruction and array.)
template <typename VT>
__global__ void __launch_bounds__(SLAB_BLOCK)
k_slab_scatter(Geometry g, const int32_t *__restrict__ row_ptr, const uint32_t *__restrict__ tile_ptr,
               const int32_t *__restrict__ col, const VT *__restrict__ val, int S, int bits, int shift,
               const uint32_t *__restrict__ chunk_start, int32_t *__restrict__ col2, VT *__restrict__ val2,
               uint32_t *__restrict__ key2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = g.tile_elems;
    VT *sval = reinterpret_cast<VT *>(smem);
    int32_t *scol = reinterpret_cast<int32_t *>(smem + (size_t)T * sizeof(VT));
    uint32_t *off = reinterpret_cast<uint32_t *>(smem + (size_t)T * (sizeof(VT) + 4));
    uint16_t *inv = reinterpret_cast<uint16_t *>(off + (size_t)(T / OMEGA) * S);
    unsigned char *skey = reinterpret_cast<unsigned char *>(inv + T);
    unsigned char *srank = skey + T;
    constexpr int ROWCAP = 1024;
    __shared__ int32_t srow[ROWCAP];
    __shared__ uint32_t slab_first[SLAB_MAX], gstart[SLAB_MAX], lstart[SLAB_MAX + 1];

    const int t = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & (OMEGA - 1), wave = tid >> 6;
    const size_t base = (size_t)t * T;
    const int E = (int)((size_t)g.nnz - base < (size_t)T ? (size_t)g.nnz - base : (size_t)T);
    const bool tr = tile_is_transposed(g, tile_ptr, t);
    // phase 0: coalesced read of the tile, LDS copy in CSR order
    for (int q = tid; q < E; q += SLAB_BLOCK) {
        const int c = tr ? (q & (OMEGA - 1)) * g.sigma + (q >> 6) : q;
        const int32_t ci = col[base + q];
        scol[c] = ci;
        sval[c] = val[base + q];
        skey[c] = (unsigned char)slab_of((uint32_t)ci, shift, bits);
    }
    if (tid < S) {
        slab_first[tid] = chunk_start[(size_t)tid * g.p]; // where slab tid begins in the child
        gstart[tid] = chunk_start[(size_t)tid * g.p + t]; // where this tile's run of slab tid begins
    }
    // the row of element j = number of row pointers in (rs, re] that are <= j: the tile's slice of row_ptr is searched in
    // LDS (a bisection of global memory per element was six dependent loads; a tile spans ~T / mean row length rows)
    const int rs = (int)(tile_ptr[t] & ROW_MASK);
    const int re = (int)(tile_ptr[t + 1] & ROW_MASK);
    const bool rows_in_lds = re - rs <= ROWCAP;
    if (rows_in_lds)
        for (int i = tid; i < re - rs; i += SLAB_BLOCK)
            srow[i] = row_ptr[rs + 1 + i];
    __syncthreads();
    const int nchunks = (E + OMEGA - 1) / OMEGA;
    // phase 1: per 64-element chunk, how many elements go to each slab, and every element's rank among them
    for (int ch = wave; ch < nchunks; ch += SLAB_BLOCK / OMEGA) {
        const int c = ch * OMEGA + lane;
        const bool valid = c < E;
        const int k = valid ? skey[c] : 0;
        if (lane < S)
            off[ch * S + lane] = 0;
        const unsigned long long same = same_slab_lanes(k, valid, bits);
        if (valid) {
            const int rank = __popcll(same & ((1ull << lane) - 1ull));
            srank[c] = (unsigned char)rank;
            if (rank == 0)
                off[ch * S + k] = (uint32_t)__popcll(same);
        }
    }
    __syncthreads();
    // phase 2: exclusive scan over the chunks for every slab (positions inside the slab's run of this tile)
    if (tid < S) {
        uint32_t run = 0;
        for (int ch = 0; ch < nchunks; ch++) {
            const uint32_t v = off[ch * S + tid];
            off[ch * S + tid] = run;
            run += v;
        }
        lstart[tid + 1] = run;
    }
    __syncthreads();
    if (tid == 0) {
        lstart[0] = 0;
        for (int k = 0; k < S; k++)
            lstart[k + 1] += lstart[k];
    }
    __syncthreads();
    // phase 3: position of every element in the slab-sorted tile
    for (int c = tid; c < E; c += SLAB_BLOCK) {
        const int k = skey[c];
        inv[lstart[k] + off[(c >> 6) * S + k] + srank[c]] = (uint16_t)c;
    }
    __syncthreads();
    // phase 4: write slab after slab
    for (int o = tid; o < E; o += SLAB_BLOCK) {
        const int c = inv[o];
        const int k = skey[c];
        const size_t dst = (size_t)gstart[k] + ((uint32_t)o - lstart[k]);
        const int j = (int)(base + c);
        const int row = rs + (rows_in_lds ? upper_bound_i32(srow, j, re - rs) : upper_bound_i32(row_ptr + rs + 1, j, re - rs));
        col2[dst] = scol[c];
        val2[dst] = sval[c];
        key2[dst] = (uint32_t)row | (dst == slab_first[k] ? SLAB_KEY_FIRST : 0u);
    }
}

// The key of a child element: its parent row, bit 31 set on the first element of every slab (one word per non-zero; the
// slab itself is implied by the position).  A segment starts where the row changes or a slab begins.
__device__ __forceinline__ bool key_starts(uint32_t a, uint32_t before)
{
    return (a & SLAB_KEY_FIRST) != 0 || ((a ^ before) & ~SLAB_KEY_FIRST) != 0;
}
// bit i = a segment starts at j + i, for the four keys at j (a multiple of 4), limited to positions below hi; k[] = the keys
__device__ __forceinline__ unsigned segment_starts4(const uint32_t *__restrict__ key, long long j, long long hi, uint32_t k[4])
{
    if (j >= hi)
        return 0u;
    const uint32_t before = j > 0 ? key[j - 1] : 0u;
    if (j + 4 <= hi) {
        const uint4 v = *reinterpret_cast<const uint4 *>(key + j);
        k[0] = v.x, k[1] = v.y, k[2] = v.z, k[3] = v.w;
    } else {
        for (int i = 0; i < 4; i++)
            k[i] = j + i < hi ? key[j + i] : 0u;
    }
    unsigned f = j == 0 || key_starts(k[0], before) ? 1u : 0u;
#pragma unroll
    for (int i = 1; i < 4; i++)
        f |= (j + i < hi && key_starts(k[i], k[i - 1]) ? 1u : 0u) << i;
    return f;
}

// Row blocks of the combine (COMBINE_ROWS rows per block = one wavefront of k_slab_combine).
constexpr int COMBINE_ROWS = 256;

// Segment starts -> row_ptr' in three small steps on a fixed partition of the keys
// into <= SEG_BLOCKS chunks: per-chunk counts (no atomics), one-workgroup scan of the counts (also the total m'), then
// every chunk re-reads its keys and writes its starts behind its offset (ballot ranks, no atomics).  rocprim::select on
// a counting iterator did this in 4.8 ms on R-MAT 24 after a 3.0-ms counting pass; the two passes here read the keys
// twice at stream speed.
constexpr int SEG_BLOCKS = 2048;
constexpr int SEG_STEP = SLAB_BLOCK * 4; // keys per workgroup and step: four consecutive ones per thread (one 16-byte load)
__host__ __device__ inline long long seg_chunk(int nnz, int blocks)
{
    long long c = ((long long)nnz + blocks - 1) / blocks;
    return (c + SEG_STEP - 1) / SEG_STEP * SEG_STEP;
}
static int seg_blocks(int nnz)
{
    long long b = ((long long)nnz + 4095) / 4096;
    return (int)(b < 1 ? 1 : (b > SEG_BLOCKS ? SEG_BLOCKS : b));
}

__global__ void __launch_bounds__(SLAB_BLOCK)
k_slab_count_segments(int nnz, const uint32_t *__restrict__ key, unsigned int *__restrict__ block_count)
{
    __shared__ unsigned part[SLAB_BLOCK / OMEGA];
    const long long chunk = seg_chunk(nnz, gridDim.x);
    const long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk < nnz ? lo + chunk : nnz;
    unsigned local = 0;
    for (long long j = lo + (long long)threadIdx.x * 4; j < hi; j += SEG_STEP) {
        uint32_t k[4];
        local += (unsigned)__popc(segment_starts4(key, j, hi, k));
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        local += __shfl_xor(local, d, OMEGA);
    if ((threadIdx.x & (OMEGA - 1)) == 0)
        part[threadIdx.x / OMEGA] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned total = 0;
        for (int w = 0; w < SLAB_BLOCK / OMEGA; w++)
            total += part[w];
        block_count[blockIdx.x] = total;
    }
}

// exclusive scan of the <= SEG_BLOCKS chunk counts in place (one workgroup); block_count[blocks] and *total = m'
__global__ void __launch_bounds__(1024) k_slab_scan_counts(int blocks, unsigned int *__restrict__ block_count,
                                                           unsigned int *__restrict__ total)
{
    __shared__ unsigned wave_total[16];
    const int per = (blocks + 1023) / 1024; // 1 or 2
    const int first = (int)threadIdx.x * per;
    unsigned v[2] = {0u, 0u};
    for (int i = 0; i < per; i++)
        if (first + i < blocks)
            v[i] = block_count[first + i];
    const unsigned sum = v[0] + v[1];
    const int lane = threadIdx.x & (OMEGA - 1), w = threadIdx.x >> 6;
    unsigned incl = sum;
#pragma unroll
    for (int d = 1; d < OMEGA; d <<= 1) {
        const unsigned o = __shfl_up(incl, d, OMEGA);
        if (lane >= d)
            incl += o;
    }
    if (lane == OMEGA - 1)
        wave_total[w] = incl;
    __syncthreads();
    unsigned run = incl - sum;
    for (int k = 0; k < w; k++)
        run += wave_total[k];
    for (int i = 0; i < per; i++)
        if (first + i < blocks) {
            block_count[first + i] = run;
            run += v[i];
        }
    if (threadIdx.x == 1023) {
        block_count[blocks] = run;
        *total = run;
    }
}

__global__ void __launch_bounds__(SLAB_BLOCK)
k_slab_emit_segments(int nnz, const uint32_t *__restrict__ key, const unsigned int *__restrict__ block_offset,
                     int32_t *__restrict__ row_ptr2, unsigned char *__restrict__ rowidx)
{
    __shared__ unsigned wave_count[SLAB_BLOCK / OMEGA];
    const long long chunk = seg_chunk(nnz, gridDim.x);
    const long long lo = (long long)blockIdx.x * chunk, hi = lo + chunk < nnz ? lo + chunk : nnz;
    const int lane = threadIdx.x & (OMEGA - 1), w = threadIdx.x / OMEGA;
    unsigned out = block_offset[blockIdx.x];
    for (long long j0 = lo; j0 < hi; j0 += SEG_STEP) { // (chunk is a multiple of the step: uniform trip count)
        const long long j = j0 + (long long)threadIdx.x * 4;
        uint32_t k4[4];
        const unsigned f = segment_starts4(key, j, hi, k4);
        const unsigned own = (unsigned)__popc(f);
        unsigned incl = own;
#pragma unroll
        for (int d = 1; d < OMEGA; d <<= 1) {
            const unsigned o = __shfl_up(incl, d, OMEGA);
            if (lane >= d)
                incl += o;
        }
        if (lane == OMEGA - 1)
            wave_count[w] = incl;
        __syncthreads();
        unsigned before = 0, all = 0;
        for (int k = 0; k < SLAB_BLOCK / OMEGA; k++) {
            before += k < w ? wave_count[k] : 0u;
            all += wave_count[k];
        }
        unsigned at = out + before + incl - own;
#pragma unroll
        for (int i = 0; i < 4; i++)
            if (f & (1u << i)) {
                row_ptr2[at] = (int32_t)(j + i);
                rowidx[at++] = (unsigned char)(k4[i] & (COMBINE_ROWS - 1)); // the segment's row inside its row block
            }
        out += all;
        __syncthreads();
    }
}

// base[b * S + k] = first segment whose (slab, row block) is >= (k, b): the segments are sorted by (slab, row), so the
// partials of (block b, slab k) are P[base[b][k] .. base[b + 1][k]); entry b = nblk of slab k is the first segment of
// slab k + 1.  One thread per entry: the workgroup first finds the segment range of every slab (the first segment at or
// after the slab's first element), then every thread bisects its slab's range by row (no atomics, no serial gap filling).
__global__ void __launch_bounds__(SLAB_BLOCK)
k_slab_base(int m2, int nnz, int S, int nblk, const int32_t *__restrict__ row_ptr2, const uint32_t *__restrict__ key,
            const uint32_t *__restrict__ chunk_start, int p, uint32_t *__restrict__ base, int32_t *__restrict__ row_ptr2_end)
{
    __shared__ int slab_seg[SLAB_MAX + 1];
    if (blockIdx.x == 0 && threadIdx.x == 0)
        *row_ptr2_end = nnz; // row_ptr'[m'] (the segment pass wrote the m' starts)
    if ((int)threadIdx.x <= S) {
        const uint32_t first = (int)threadIdx.x < S ? chunk_start[(size_t)threadIdx.x * p] : (uint32_t)nnz;
        int lo = 0, hi = m2;
        while (lo < hi) {
            const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
            if ((uint32_t)row_ptr2[mid] < first)
                lo = mid + 1;
            else
                hi = mid;
        }
        slab_seg[threadIdx.x] = lo;
    }
    __syncthreads();
    const long long e = (long long)blockIdx.x * SLAB_BLOCK + threadIdx.x;
    if (e >= (long long)(nblk + 1) * S)
        return;
    const int b = (int)(e / S), k = (int)(e % S);
    const uint32_t want = (uint32_t)b * COMBINE_ROWS; // (a block index of nblk is beyond every row)
    int lo = slab_seg[k], hi = slab_seg[k + 1];
    while (lo < hi) {
        const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
        if ((key[row_ptr2[mid]] & ~SLAB_KEY_FIRST) < want)
            lo = mid + 1;
        else
            hi = mid;
    }
    base[e] = (uint32_t)lo;
}

// one bit per row of the parent: the row owns a non-zero (rows without one are not written by SpMV)
__global__ void __launch_bounds__(SLAB_BLOCK)
k_slab_nonempty(int m, const int32_t *__restrict__ row_ptr, uint32_t *__restrict__ bits)
{
    const int r = blockIdx.x * SLAB_BLOCK + threadIdx.x;
    const bool on = r < m && row_ptr[r + 1] > row_ptr[r];
    const unsigned long long b = __ballot(on);
    const int lane = threadIdx.x & (OMEGA - 1);
    if ((lane & 31) == 0 && r < m + 32)
        bits[r >> 5] = (uint32_t)(b >> lane);
}

// y[r] = sum of the partials of row r.  One wavefront per COMBINE_ROWS consecutive rows.  First round trip: the block's
// 2 S run bounds (one coalesced load) and its non-empty bits; then, round after round, the next 64 partials and row
// bytes of EVERY run are requested together (lanes beyond a run's end re-read its last element: no branch, no extra
// line) and added into the block's accumulators in LDS, slab after slab -- inside one wavefront the LDS operations are
// ordered, inside a run every row occurs once, and masked lanes add into a dummy slot of their own, so the whole round
// is straight-line code.  A row's partials are therefore added in the fixed order (round, slab): deterministic, no
// floating-point atomics.  Most blocks need one round; a block whose 256 rows all own a segment in some slab needs four.
// (Round 2 gave a wavefront 64 rows and fetched the partials row-wise: one 15 %-full load instruction per (block, slab),
// 4.2 M of them on R-MAT 24, and 16 ballots per block to index them.)
#if defined(CSR5_COMBINE_STAMPS) // experiment builds only: wall-clock stamps (100 MHz) of every wavefront's start and end
__device__ unsigned long long g_combine_stamps[2 * (1 << 17)];
__device__ unsigned long long g_combine_phase[4 * (1 << 17)]; // after the bounds arrived, after the partials arrived, after the adds, end
#define COMBINE_PHASE(i)                                                                                               \
    do {                                                                                                               \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");                                                    \
        if (lane == 0 && blk < (1 << 17))                                                                              \
            g_combine_phase[4 * blk + (i)] = wall_clock64();                                                           \
    } while (0)
extern "C" int csr5hip_debug_combine_phase(unsigned long long *dst, int count)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_combine_phase), (size_t)count * sizeof(unsigned long long));
}
extern "C" int csr5hip_debug_combine_stamps(unsigned long long *dst, int count)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_combine_stamps), (size_t)count * sizeof(unsigned long long));
}
#else
#define COMBINE_PHASE(i)
#endif

#ifndef CSR5_COMBINE_WAVES
#define CSR5_COMBINE_WAVES 4
#endif
constexpr int COMBINE_WAVES = CSR5_COMBINE_WAVES; // row blocks (= wavefronts) per workgroup of the combine
// one 256-row block `blk` by one wavefront; `acc` = its COMBINE_ROWS + OMEGA accumulators in LDS
template <typename VT, int S>
__device__ __forceinline__ void combine_block(int blk, int lane, VT *acc, int m, int tail_start, int zero_empty, int m2,
                                              const uint32_t *__restrict__ base, const unsigned char *__restrict__ rowidx,
                                              const uint32_t *__restrict__ nonempty, const VT *__restrict__ P,
                                              VT *__restrict__ y)
{
    const int r0 = blk * COMBINE_ROWS;
    // run bounds of the block: words [blk * S, blk * S + 2 S) = this block's and the next block's starts
    constexpr int BW = (2 * S + OMEGA - 1) / OMEGA;
    uint32_t bw[BW];
#pragma unroll
    for (int i = 0; i < BW; i++) {
        const int e = i * OMEGA + lane;
        bw[i] = base[(size_t)blk * S + (e < 2 * S ? e : 0)];
    }
    uint32_t ne[COMBINE_ROWS / OMEGA];
#pragma unroll
    for (int j = 0; j < COMBINE_ROWS / OMEGA; j++)
        ne[j] = nonempty[(r0 >> 5) + 2 * j + (lane >> 5)];
#pragma unroll
    for (int j = 0; j < COMBINE_ROWS / OMEGA; j++)
        acc[j * OMEGA + lane] = 0;
    COMBINE_PHASE(0);
    auto bound = [&](int e) -> int { return __builtin_amdgcn_readlane((int)bw[e / OMEGA], e % OMEGA); };
    constexpr int G = S < 16 ? S : 16; // runs whose loads are in flight together
    const unsigned dummy = COMBINE_ROWS + lane;
#pragma unroll
    for (int k0 = 0; k0 < S; k0 += G) {
        int lo[G], len[G], longest = 0;
#pragma unroll
        for (int q = 0; q < G; q++) {
            lo[q] = bound(k0 + q);
            len[q] = bound(S + k0 + q) - lo[q];
            longest = len[q] > longest ? len[q] : longest;
        }
        for (int off = 0; off < longest; off += OMEGA) {
            VT part[G];
            unsigned idx[G];
#pragma unroll
            for (int q = 0; q < G; q++) {
                // element off + lane of run q, or -- beyond its end -- its last one (an empty run reads its neighbour's)
                int j = off + lane < len[q] ? off + lane : len[q] - 1;
                j += lo[q];
                j = j < 0 ? 0 : (j < m2 ? j : m2 - 1);
                part[q] = P[j];
                idx[q] = rowidx[j];
            }
            COMBINE_PHASE(1);
#pragma unroll
            for (int q = 0; q < G; q++) {
                const unsigned slot = off + lane < len[q] ? idx[q] : dummy;
                acc[slot] += part[q];
            }
        }
    }
    COMBINE_PHASE(2);
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < COMBINE_ROWS / OMEGA; j++) {
        const int r = r0 + j * OMEGA + lane;
        if (r < m) {
            if ((ne[j] >> (lane & 31)) & 1u)
                y[r] = acc[j * OMEGA + lane];
            else if (zero_empty || r >= tail_start)
                y[r] = 0;
        }
    }
}

template <typename VT, int S>
__global__ void __launch_bounds__(COMBINE_WAVES * OMEGA)
k_slab_combine(int m, int tail_start, int zero_empty, int m2, const uint32_t *__restrict__ base,
               const unsigned char *__restrict__ rowidx, const uint32_t *__restrict__ nonempty,
               const VT *__restrict__ P, VT *__restrict__ y)
{
    __shared__ VT acc_all[COMBINE_WAVES][COMBINE_ROWS + OMEGA]; // + one dummy slot per lane
    const int lane = threadIdx.x & (OMEGA - 1), w = threadIdx.x >> 6;
    // Which 1 024 rows does this workgroup take?  Workgroup b runs on XCD b % 8 (dispatch order), so "rows 1024 b ..." hands
    // every XCD the row groups with ONE pattern of bits 10-12 of the row index -- and on a graph whose row lengths follow the
    // bits of the index (R-MAT: a row with a zero bit is 3.2x as long as its sibling with a one) XCD 0 gets 31x the partials of
    // XCD 7: the kernel ran 180 us with 13 of 32 wavefronts resident and a 40-us tail of one XCD (wall-clock stamps per
    // wavefront, scripts/experiments/round5/combine_stamps.py).  Inside every octet of consecutive workgroups the row groups
    // are therefore rotated by a hash of the octet's number: every XCD sees every bit pattern equally often, and which one it
    // sees is unrelated to the octet's own (equally skewed) index bits.
    int wg = blockIdx.x;
    {
        const int octet = wg >> 3;
        if ((octet << 3) + 8 <= (int)gridDim.x)
            wg = (octet << 3) + (((wg & 7) + (int)(((unsigned)octet * 0x9E3779B1u) >> 29)) & 7);
    }
    const int blk = wg * COMBINE_WAVES + w;
    const int r0 = blk * COMBINE_ROWS;
    if (r0 >= m)
        return;
#if defined(CSR5_COMBINE_STAMPS)
    if (lane == 0 && blk < (1 << 17))
        g_combine_stamps[2 * blk] = wall_clock64();
#endif
    combine_block<VT, S>(blk, lane, acc_all[w], m, tail_start, zero_empty, m2, base, rowidx, nonempty, P, y);
#if defined(CSR5_COMBINE_STAMPS)
    if (lane == 0 && blk < (1 << 17))
        g_combine_stamps[2 * blk + 1] = wall_clock64();
#endif
}


// Persistent form (CSR5_COMBINE_PERSISTENT builds): 8 workgroups per CU stay resident and every wavefront draws row blocks from
// a ticket counter (the next ticket is requested before the current block is worked on, so its round trip is hidden).  The
// counter re-arms itself: every wavefront draws exactly one ticket beyond the last block, so the wavefront that draws ticket
// nblk + nwaves - 1 is the last to touch it and stores 0.  Which wavefront adds a block does not change a bit of y.
template <typename VT, int S>
__global__ void __launch_bounds__(COMBINE_WAVES * OMEGA)
k_slab_combine_persistent(int m, int tail_start, int zero_empty, int m2, const uint32_t *__restrict__ base,
                          const unsigned char *__restrict__ rowidx, const uint32_t *__restrict__ nonempty,
                          const VT *__restrict__ P, VT *__restrict__ y, unsigned *__restrict__ ticket)
{
    __shared__ VT acc_all[COMBINE_WAVES][COMBINE_ROWS + OMEGA];
    const int lane = threadIdx.x & (OMEGA - 1), w = threadIdx.x >> 6;
    const unsigned nblk = (unsigned)((m + COMBINE_ROWS - 1) / COMBINE_ROWS), nwaves = gridDim.x * COMBINE_WAVES;
#if defined(CSR5_COMBINE_STATIC)
    // static schedule: round k hands wavefront g block k * nwaves + ((g + hash(k)) mod nwaves): the blocks of one wavefront come
    // from different parts of the row range AND carry unrelated low index bits
    const unsigned g = blockIdx.x * COMBINE_WAVES + w;
    for (unsigned k = 0; k * nwaves < nblk; k++) {
        const unsigned t = k * nwaves + (g + (k * 0x9E3779B1u >> 8)) % nwaves;
        if (t < nblk) {
            combine_block<VT, S>((int)t, lane, acc_all[w], m, tail_start, zero_empty, m2, base, rowidx, nonempty, P, y);
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            __builtin_amdgcn_wave_barrier();
        }
    }
    (void)ticket;
    return;
#endif
    auto draw = [&]() -> unsigned {
        unsigned t = 0;
        if (lane == 0)
            t = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return (unsigned)__builtin_amdgcn_readfirstlane((int)t);
    };
    unsigned t = draw();
    while (t < nblk) {
        const unsigned next = draw();
        combine_block<VT, S>((int)t, lane, acc_all[w], m, tail_start, zero_empty, m2, base, rowidx, nonempty, P, y);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // the next block's LDS writes stay behind this block's reads
        __builtin_amdgcn_wave_barrier();
        t = next;
    }
    if (t == nblk + nwaves - 1u && lane == 0)
        __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// ---- hot columns of every slab (LDS table of the persistent kernel k_spmv_range, csr5_hot.hip) ---------------------
// Power-law inputs concentrate their non-zeros on few columns: the HOT_CAPACITY most used columns of a slab (those
// used at least `min_count` times -- a table entry is staged once per workgroup and SpMV, so a rarely used column
// would cost more than it saves) get a slot in the slab's table and their column words are rewritten to
// 0x80000000 | slot in the child's column_index.  Selection: per-slab histogram of the column use counts, threshold =
// the smallest count whose columns all fit, the remaining room goes to columns of the next lower count.
constexpr int HOT_BUCKETS = 2048;

// Use counts of the columns from a SAMPLE of the non-zeros: one 64-element chunk out of every `stride` (the reads stay
// coalesced).  The table only needs to know which columns are popular, and a popular column is exactly what makes a
// count slow: increments of ONE word are served one after the other, ~73 ns each (R-MAT 24 sends 370 k of them to its top
// column: 27 ms for a full count, and still 0.83 ms = 11.5 k x 73 ns for the 1/32 sample).  So a workgroup first merges
// its samples in a small direct-mapped table in LDS -- a column that finds its slot free or its own adds there, any
// other goes to memory at once -- and flushes every used slot with one add: the top column then costs one add per
// workgroup.  (Rows of unpopular columns fill the table too; a popular one shows up early enough to find its slot.)
constexpr int COUNT_SLOTS = 16384, COUNT_BLOCK = 1024, COUNT_WGS = 256, COUNT_AHEAD = 4;
__global__ void __launch_bounds__(COUNT_BLOCK)
k_col_count(int nnz, int stride, const int32_t *__restrict__ col, uint32_t *__restrict__ cnt)
{
    extern __shared__ uint32_t count_lds[];
    uint32_t *key = count_lds, *hits = count_lds + COUNT_SLOTS;
    for (int i = threadIdx.x; i < COUNT_SLOTS; i += COUNT_BLOCK) {
        key[i] = 0xFFFFFFFFu;
        hits[i] = 0;
    }
    __syncthreads();
    const size_t chunks = ((size_t)nnz + OMEGA - 1) / OMEGA;
    const size_t sampled = (chunks + stride - 1) / stride;
    const int lane = threadIdx.x & (OMEGA - 1);
    const size_t step = (size_t)gridDim.x * (COUNT_BLOCK / OMEGA);
    for (size_t w0 = ((size_t)blockIdx.x * COUNT_BLOCK + threadIdx.x) / OMEGA; w0 < sampled; w0 += step * COUNT_AHEAD) {
        uint32_t c[COUNT_AHEAD];
#pragma unroll
        for (int u = 0; u < COUNT_AHEAD; u++) {
            const size_t w = w0 + u * step;
            const size_t i = w * stride * OMEGA + lane;
            c[u] = w < sampled && i < (size_t)nnz ? (uint32_t)col[i] : 0xFFFFFFFFu;
        }
#pragma unroll
        for (int u = 0; u < COUNT_AHEAD; u++) {
            if (c[u] == 0xFFFFFFFFu)
                continue;
            const uint32_t slot = (c[u] * 2654435761u) >> 18; // 14 bits
            const uint32_t owner = atomicCAS(&key[slot], 0xFFFFFFFFu, c[u]);
            if (owner == 0xFFFFFFFFu || owner == c[u])
                atomicAdd(&hits[slot], 1u);
            else
                atomicAdd(&cnt[c[u]], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < COUNT_SLOTS; i += COUNT_BLOCK)
        if (hits[i])
            atomicAdd(&cnt[key[i]], hits[i]);
}

// Histogram of the use counts per slab.  Most columns have SMALL counts (on a matrix without popular columns all of
// them: one million increments on eighty words took 0.56 ms), so the buckets below HOT_LOW are accumulated per workgroup
// in LDS and flushed once; only the rare large counts go to memory directly.
constexpr int HOT_LOW = 64;
constexpr int HOT_HIST_COLS = 4096; // columns per workgroup
__global__ void __launch_bounds__(SLAB_BLOCK)
k_hot_hist(int n, const uint32_t *__restrict__ cnt, int S, int bits, int shift, uint32_t *__restrict__ chist)
{
    __shared__ uint32_t low[SLAB_MAX * HOT_LOW];
    for (int i = threadIdx.x; i < S * HOT_LOW; i += SLAB_BLOCK)
        low[i] = 0;
    __syncthreads();
    const int first = blockIdx.x * HOT_HIST_COLS;
    for (int c = first + threadIdx.x; c < first + HOT_HIST_COLS && c < n; c += SLAB_BLOCK) {
        const uint32_t v = cnt[c];
        if (!v)
            continue;
        const uint32_t k = slab_of((uint32_t)c, shift, bits);
        if (v < HOT_LOW)
            atomicAdd(&low[k * HOT_LOW + v], 1u);
        else
            atomicAdd(&chist[(size_t)k * HOT_BUCKETS + (v < HOT_BUCKETS - 1 ? v : HOT_BUCKETS - 1)], 1u);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < S * HOT_LOW; i += SLAB_BLOCK)
        if (low[i])
            atomicAdd(&chist[(size_t)(i / HOT_LOW) * HOT_BUCKETS + i % HOT_LOW], low[i]);
}

// one wavefront per slab: thr[k] = smallest bucket b >= min_count such that the columns with count >= b fit, i.e.
// suffix(b) = sum of h[b..] <= capacity - 1 (slot 0 is reserved).  suffix() only grows as b falls, so the answer is
// the minimum over the lanes of "lowest fitting bucket of my 32" (a serial walk of 2 048 dependent loads took 0.12 ms).
__global__ void __launch_bounds__(OMEGA)
k_hot_threshold(int capacity, int min_count, const uint32_t *__restrict__ chist, uint32_t *__restrict__ thr)
{
    constexpr int PER = HOT_BUCKETS / OMEGA;
    const int lane = threadIdx.x;
    const uint32_t *h = chist + (size_t)blockIdx.x * HOT_BUCKETS + lane * PER;
    uint32_t v[PER];
    unsigned long long own = 0;
#pragma unroll
    for (int i = 0; i < PER; i++) {
        v[i] = h[i];
        own += v[i];
    }
    // exclusive suffix over the lanes: sum of the lanes above
    unsigned long long incl = own;
#pragma unroll
    for (int d = 1; d < OMEGA; d <<= 1) {
        const unsigned long long o = __shfl_down(incl, d, OMEGA);
        if (lane + d < OMEGA)
            incl += o;
    }
    unsigned long long total = incl - own;
    int best = HOT_BUCKETS; // suffix(HOT_BUCKETS) = 0 always fits
#pragma unroll
    for (int i = PER - 1; i >= 0; i--) {
        total += v[i];
        const int b = lane * PER + i;
        if (b >= min_count && total <= (unsigned long long)(capacity - 1))
            best = b;
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const int o = __shfl_xor(best, d, OMEGA);
        best = o < best ? o : best;
    }
    // room = slots left for the columns of the next lower count: capacity - 1 - suffix(best)
    unsigned long long at_best = 0; // suffix(best), held by the lane that owns bucket `best` (0 for best = HOT_BUCKETS)
    {
        unsigned long long run = incl - own;
#pragma unroll
        for (int i = PER - 1; i >= 0; i--) {
            run += v[i];
            if (lane * PER + i == best)
                at_best = run;
        }
    }
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        at_best += __shfl_xor(at_best, d, OMEGA);
    if (lane == 0) {
        thr[blockIdx.x] = (uint32_t)best;
        thr[gridDim.x + blockIdx.x] = (uint32_t)((unsigned long long)(capacity - 1) - at_best);
    }
}

// Every column with count >= thr is marked hot; columns of the next lower count fill the room that is left (thr[S + k],
// from k_hot_threshold), first come first served.  A workgroup walks HOT_ASSIGN_COLS columns twice: first it counts the
// second-class columns it wants per slab (LDS), reserves them with ONE global add per slab, then hands them out from LDS
// counters (196 k adds on 16 words -- one per chosen column -- took 0.9 ms on R-MAT 24).  The bitmap only says WHETHER a
// column gets a slot; k_hot_rank numbers them.
constexpr int HOT_ASSIGN_COLS = 16384;
__global__ void __launch_bounds__(SLAB_BLOCK)
k_hot_assign(int n, int S, const uint32_t *__restrict__ cnt, int bits, int shift, const uint32_t *__restrict__ thr,
             int min_count, int32_t *__restrict__ fill, uint32_t *__restrict__ hotbits, size_t G,
             unsigned long long *__restrict__ covered)
{
    __shared__ int want[SLAB_MAX], base[SLAB_MAX], given[SLAB_MAX];
    if (threadIdx.x < SLAB_MAX)
        want[threadIdx.x] = given[threadIdx.x] = base[threadIdx.x] = 0;
    __syncthreads();
    const int first = blockIdx.x * HOT_ASSIGN_COLS;
    const int last = first + HOT_ASSIGN_COLS < n ? first + HOT_ASSIGN_COLS : n;
    // 0 = not chosen, 1 = first class (count >= thr), 2 = second class (count == thr - 1)
    auto chosen = [&](int c, uint32_t &k, uint32_t &v) -> int {
        v = cnt[c];
        if (!v)
            return 0;
        k = slab_of((uint32_t)c, shift, bits);
        const uint32_t bucket = v < HOT_BUCKETS - 1 ? v : HOT_BUCKETS - 1;
        const uint32_t b = thr[k];
        return bucket >= b ? 1 : (bucket + 1 == b && (int)bucket >= min_count ? 2 : 0);
    };
    auto mark = [&](int c, uint32_t k) {
        const uint32_t local = slab_local((uint32_t)c, shift, bits);
        atomicOr(&hotbits[(size_t)k * G * 4 + (local >> 5)], 1u << (local & 31));
    };
    uint32_t k = 0, v = 0;
    unsigned long long got = 0;
    for (int c = first + (int)threadIdx.x; c < last; c += SLAB_BLOCK) {
        const int cls = chosen(c, k, v);
        if (cls == 1) {
            mark(c, k);
            got += v;
        } else if (cls == 2) {
            atomicAdd(&want[k], 1);
        }
    }
    __syncthreads();
    if (threadIdx.x < SLAB_MAX && want[threadIdx.x])
        base[threadIdx.x] = atomicAdd(&fill[threadIdx.x], want[threadIdx.x]);
    __syncthreads();
    for (int c = first + (int)threadIdx.x; c < last; c += SLAB_BLOCK)
        if (chosen(c, k, v) == 2 && want[k] && base[k] + atomicAdd(&given[k], 1) < (int)thr[S + k]) {
            mark(c, k);
            got += v;
        }
    // sampled non-zeros that found a slot: one increment per wavefront
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        got += __shfl_xor(got, d, OMEGA);
    if ((threadIdx.x & (OMEGA - 1)) == 0 && got)
        atomicAdd(covered, got);
}

// clamp the slot counts; first tile OWNED by every slab (a tile belongs to the slab of its first element)
__global__ void __launch_bounds__(SLAB_MAX + 1)
k_hot_finish(int S, int p_hist, int p, int T, int nnz, int capacity, const uint32_t *__restrict__ chunk_start,
             int32_t *__restrict__ hot_count, int32_t *__restrict__ tile0, int32_t *__restrict__ slab_off)
{
    // p_hist = tiles of the PARENT (row stride of the chunk table), p / T = tiles / tile size of the CHILD
    const int k = threadIdx.x;
    if (k > S)
        return;
    const long long off = k < S ? (long long)chunk_start[(size_t)k * p_hist] : (long long)nnz;
    slab_off[k] = (int32_t)off;
    long long t = (off + T - 1) / T;
    tile0[k] = (int32_t)(t < p - 1 ? t : p - 1);
    if (k < S && hot_count[k] > capacity)
        hot_count[k] = capacity;
}

// Slots in column order: pre[g] = slot of the first hot column of group g (slot 0 is reserved, so the count starts at 1),
// hot_cols[slot] = column, hot_count[k] = slots in use.  Workgroup (k, j) numbers the groups [j, j + 1) * RANK_BLOCK of
// slab k; it first counts the set bits in front of them itself (at most 128 KB of bitmap) instead of waiting for the
// workgroups before it.
constexpr int RANK_BLOCK = 1024;
__global__ void __launch_bounds__(RANK_BLOCK)
k_hot_rank(int bits, int shift, int capacity, size_t G, const uint4 *__restrict__ hotbits, uint16_t *__restrict__ pre,
           int32_t *__restrict__ hot_cols, int32_t *__restrict__ hot_count)
{
    __shared__ uint32_t wave_total[RANK_BLOCK / OMEGA];
    const uint32_t k = blockIdx.x;
    const size_t g0 = (size_t)blockIdx.y * RANK_BLOCK;
    const int lane = threadIdx.x & (OMEGA - 1), w = threadIdx.x / OMEGA;
    auto bits_of = [](const uint4 v) { return (uint32_t)(__popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w)); };
    auto block_scan = [&](uint32_t own, uint32_t &before, uint32_t &all) { // exclusive position of `own`, and the total
        uint32_t incl = own;
#pragma unroll
        for (int d = 1; d < OMEGA; d <<= 1) {
            const uint32_t o = __shfl_up(incl, d, OMEGA);
            if (lane >= d)
                incl += o;
        }
        __syncthreads(); // (wave_total is reused)
        if (lane == OMEGA - 1)
            wave_total[w] = incl;
        __syncthreads();
        before = incl - own;
        all = 0;
        for (int i = 0; i < RANK_BLOCK / OMEGA; i++) {
            before += i < w ? wave_total[i] : 0u;
            all += wave_total[i];
        }
    };
    uint32_t front = 0;
    for (size_t g = threadIdx.x; g < g0; g += RANK_BLOCK)
        front += bits_of(hotbits[(size_t)k * G + g]);
    uint32_t unused, running;
    block_scan(front, unused, running);
    running += 1; // slot 0 is reserved
    const size_t g = g0 + threadIdx.x;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (g < G)
        v = hotbits[(size_t)k * G + g];
    uint32_t before, all;
    block_scan(bits_of(v), before, all);
    uint32_t slot = running + before;
    if (g < G) {
        pre[(size_t)k * G + g] = (uint16_t)slot;
        const uint32_t word[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t m = word[i];
            while (m) {
                const uint32_t b = (uint32_t)__builtin_ctz(m);
                m &= m - 1;
                if ((int)slot < capacity)
                    hot_cols[(size_t)k * capacity + slot] = (int32_t)slab_column(k, (uint32_t)g * 128u + i * 32u + b, shift, bits);
                slot++;
            }
        }
    }
    if (blockIdx.y == gridDim.y - 1 && threadIdx.x == 0)
        hot_count[k] = (int32_t)(running + all);
}

// One persistent pass over the child's column words, slab after slab: workgroup b runs on XCD b % 8 (observed placement,
// used for locality only) and takes the slabs k = xcd, xcd + 8, ...; per slab it stages the slab's hot bitmap and group
// prefixes in LDS (IN_LDS; a slab with more than ~1.1 M columns reads them from memory instead) and streams its share
// of the slab's column words, four per lane and load.  What it does with an element depends on MODE:
//   ENC_MARK     marks every column that is gathered COLD (not through the table) somewhere -- one byte per column,
//                ref[slab][local id], plain stores (all writers store 1).  Counting the cold uses exactly would take one
//                global atomic per cold element: 91 M on R-MAT 24, 4.2 ms at the chip's 26.7 atomics per ns
//                (profiles/r03_probes.txt) -- the ranking below takes the popularity from the sampled counts of the table
//                selection instead and only needs to know WHICH columns to rank
//   ENC_PACK     every column word of the tiles 0 .. p-2 is written as a 24-bit code -- low 16 bits to col_lo, high 8 bits
//                to col_hi, both in the child's CSR order, which is already the order lane l of k_spmv_range wants (its
//                sigma elements are consecutive there): bit 23 = table slot in bits 0..14, else the RANK of the column in
//                its slab's frequency order (rank_of[slab][local id], < 2^22) = its index in the slab's cold region of
//                the permuted copy of x (csr5_hot.hip); bit 22 = the element starts a row of the child (the bit flag of the
//                reference's descriptor, from the segment keys).  3 bytes per non-zero instead of 4 in the SpMV's streams; col2
//                stays plain (the CSR tail tile reads it).
// A tile belongs to the slab of its first element.  Elements behind the slab's end inside its last tile belong to the
// following slab(s): they are cold there (that tile runs with THIS slab's table) and are counted / coded with the
// numbering of THEIR slab.
constexpr int ENCODE_BLOCK = 1024, ENCODE_WGS_PER_XCD = 32, ENCODE_UNROLL = 4;
constexpr int ENC_MARK = 1, ENC_PACK = 2;
template <bool IN_LDS, int MODE>
__global__ void __launch_bounds__(ENCODE_BLOCK)
k_hot_encode(int nnz, int T, int p, int S, int bits, int shift, const int32_t *__restrict__ slab_off,
             const uint4 *__restrict__ hotbits, const uint16_t *__restrict__ hotpre, size_t G, int32_t *__restrict__ col2,
             uint16_t *__restrict__ col_lo, uint8_t *__restrict__ col_hi, uint8_t *__restrict__ ref,
             const uint32_t *__restrict__ rank_of, size_t L, const uint32_t *__restrict__ key)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    uint4 *lbits = reinterpret_cast<uint4 *>(smem);
    uint16_t *lpre = reinterpret_cast<uint16_t *>(smem + G * sizeof(uint4));
    __shared__ int32_t soff[SLAB_MAX + 1];
    if ((int)threadIdx.x <= S)
        soff[threadIdx.x] = slab_off[threadIdx.x];
    __syncthreads();
    const int xcd = blockIdx.x % NUM_XCD, j = blockIdx.x / NUM_XCD, nj = gridDim.x / NUM_XCD;
    const long long last = (long long)(p - 1) * T; // the CSR tail is processed from CSR: plain column words
    for (int mine = xcd; mine < S; mine += NUM_XCD) {
        const uint4 *gb = hotbits + (size_t)mine * G;
        const uint16_t *gp = hotpre + (size_t)mine * G;
        if (IN_LDS) {
            __syncthreads(); // (the previous slab's look-ups are done)
            for (size_t g = threadIdx.x; g < G; g += ENCODE_BLOCK) {
                lbits[g] = gb[g];
                if (MODE != ENC_MARK)
                    lpre[g] = gp[g];
            }
            __syncthreads();
        }
        const long long own_end = soff[mine + 1];
        // (the element's slot in its slab's table, or -1)
        auto slot_of = [&](uint32_t local) -> int {
            const uint32_t g = local >> 7;
            uint32_t below;
            const bool hot = hot_lookup(IN_LDS ? lbits[g] : gb[g], local & 127u, below);
            if (MODE == ENC_MARK)
                return hot ? 0 : -1;
            return hot ? (int)((uint32_t)(IN_LDS ? lpre[g] : gp[g]) + below) : -1;
        };
        auto encode = [&](int32_t c, long long pos) -> int32_t {
            const uint32_t local = slab_local((uint32_t)c, shift, bits);
            const int slot = pos < own_end ? slot_of(local) : -1;
            if (slot >= 0)
                return (int32_t)(0x800000u | (uint32_t)slot);
            const size_t at = (size_t)(pos < own_end ? (uint32_t)mine : slab_of((uint32_t)c, shift, bits)) * L + local;
            if (MODE == ENC_MARK) {
                ref[at] = 1;
                return 0;
            }
            return (int32_t)rank_of[at];
        };
        // An element inside a tile owned by the PREVIOUS slab is gathered with that slab's table in LDS: it is that
        // slab's business.  So the slab's share starts at its first own tile and covers its whole last tile, foreign
        // elements included.
        const long long begin = ((long long)soff[mine] + T - 1) / T * T;
        long long end = (own_end + T - 1) / T * T;
        end = end < last ? end : last;
        if (begin >= end)
            continue;
        int4 *c4 = reinterpret_cast<int4 *>(col2 + begin); // (T is a multiple of 64: 16-byte aligned)
        uint2 *lo4 = reinterpret_cast<uint2 *>(col_lo + begin);
        uint32_t *hi4 = reinterpret_cast<uint32_t *>(col_hi + begin);
        const long long quads = (end - begin) / 4;
        for (long long q0 = (long long)j * ENCODE_BLOCK * ENCODE_UNROLL; q0 < quads;
             q0 += (long long)nj * ENCODE_BLOCK * ENCODE_UNROLL) {
            int4 v[ENCODE_UNROLL];
#pragma unroll
            for (int u = 0; u < ENCODE_UNROLL; u++) {
                const long long q = q0 + u * ENCODE_BLOCK + threadIdx.x;
                if (q < quads)
                    v[u] = c4[q];
            }
#pragma unroll
            for (int u = 0; u < ENCODE_UNROLL; u++) {
                const long long q = q0 + u * ENCODE_BLOCK + threadIdx.x;
                if (q < quads) {
                    const long long pos = begin + q * 4;
                    v[u].x = encode(v[u].x, pos);
                    v[u].y = encode(v[u].y, pos + 1);
                    v[u].z = encode(v[u].z, pos + 2);
                    v[u].w = encode(v[u].w, pos + 3);
                    if (MODE == ENC_PACK) {
                        // bit 22 = the element starts a row of the child (a segment): the reference format's bit flag,
                        // carried by the code so that the SpMV kernel does not read the descriptor array
                        uint32_t k4[4];
                        const unsigned st = segment_starts4(key, pos, (long long)nnz, k4);
                        const uint32_t a = (uint32_t)v[u].x | ((st & 1u) << 22), b = (uint32_t)v[u].y | (((st >> 1) & 1u) << 22),
                                       c = (uint32_t)v[u].z | (((st >> 2) & 1u) << 22), d = (uint32_t)v[u].w | (((st >> 3) & 1u) << 22);
                        lo4[q] = make_uint2((a & 0xFFFFu) | (b << 16), (c & 0xFFFFu) | (d << 16));
                        hi4[q] = ((a >> 16) & 0xFFu) | (((b >> 16) & 0xFFu) << 8) | (((c >> 16) & 0xFFu) << 16) | ((d >> 16) << 24);
                    }
                }
            }
        }
        // (begin and end are multiples of the tile size, nothing is left over)
    }
}

// ---- frequency order of the cold columns of every slab (the layout of the permuted copy of x, csr5_hot.hip) ----------
// ref[slab][local id] = the column is gathered cold somewhere (ENC_MARK); cnt[column] = its sampled use count from the table
// selection (k_col_count: one 64-element chunk in `stride`).  A stable radix sort on (slab, 1023 - min(count, 1023)) with
// the index as payload puts every slab's marked columns in descending order of (sampled) use, ties -- among them all the
// columns the sample missed -- in column order; unmarked columns (only ever read through the LDS table, or by nobody) get
// the key 1024 and fall behind: the first ncold[k] positions of slab k are the slab's cold region.
constexpr uint32_t COLD_COUNT_CAP = 1023, COLD_KEY_BITS = 11;
__global__ void __launch_bounds__(SLAB_BLOCK)
k_cold_keys(size_t total, size_t L, int n, int bits, int shift, const uint8_t *__restrict__ ref, const uint32_t *__restrict__ cnt,
            uint32_t *__restrict__ keys)
{
    const size_t i = (size_t)blockIdx.x * SLAB_BLOCK + threadIdx.x;
    if (i >= total)
        return;
    const uint32_t k = (uint32_t)(i / L);
    uint32_t low = COLD_COUNT_CAP + 1;
    if (ref[i]) {
        const uint32_t col = slab_column(k, (uint32_t)(i - (size_t)k * L), shift, bits);
        const uint32_t c = col < (uint32_t)n ? cnt[col] : 0u;
        low = COLD_COUNT_CAP - (c < COLD_COUNT_CAP ? c : COLD_COUNT_CAP);
    }
    keys[i] = (k << COLD_KEY_BITS) | low;
}
// one thread per slab: ncold[k] = marked columns of slab k (their keys sort in front of (k << 11) | 1024), then
// cold_base[k] = start of the slab's cold region (exclusive prefix; cold_base[S] = all cold entries)
__global__ void __launch_bounds__(SLAB_MAX)
k_cold_layout(int S, size_t L, const uint32_t *__restrict__ keys_sorted, int32_t *__restrict__ cold_base)
{
    __shared__ int32_t n[SLAB_MAX];
    const int k = threadIdx.x;
    if (k < S) {
        const uint32_t none = ((uint32_t)k << COLD_KEY_BITS) | (COLD_COUNT_CAP + 1);
        size_t lo = (size_t)k * L, hi = lo + L;
        while (lo < hi) {
            const size_t mid = (lo + hi) >> 1;
            if (keys_sorted[mid] < none)
                lo = mid + 1;
            else
                hi = mid;
        }
        n[k] = (int32_t)(lo - (size_t)k * L);
    }
    __syncthreads();
    if (k == 0) {
        int32_t run = 0;
        for (int i = 0; i < S; i++) {
            cold_base[i] = run;
            run += n[i];
        }
        cold_base[S] = run;
    }
}
// sorted position q of slab k -> rank r = q - k L: rank_of[source index] = r, and the column of the slab's r-th cold entry
__global__ void __launch_bounds__(SLAB_BLOCK)
k_cold_rank(size_t total, size_t L, int bits, int shift, const uint32_t *__restrict__ src_sorted,
            const int32_t *__restrict__ cold_base, uint32_t *__restrict__ rank_of, int32_t *__restrict__ cold_cols)
{
    const size_t q = (size_t)blockIdx.x * SLAB_BLOCK + threadIdx.x;
    if (q >= total)
        return;
    const uint32_t k = (uint32_t)(q / L);
    const uint32_t r = (uint32_t)(q - (size_t)k * L);
    const int32_t b = cold_base[k];
    if ((int32_t)r >= cold_base[k + 1] - b)
        return;
    const uint32_t src = src_sorted[q];
    rank_of[src] = r;
    cold_cols[(size_t)b + r] = (int32_t)slab_column(k, (uint32_t)(src - (size_t)k * L), shift, bits);
}

// ---- host side -----------------------------------------------------------------------------------------
size_t slab_scatter_lds(const Geometry &g, int S, size_t vsize)
{
    const size_t T = (size_t)g.tile_elems;
    return T * (vsize + 4) + (T / OMEGA) * (size_t)S * 4 + T * 2 + T + T; // values, columns, chunk counts, inverse, slab, rank
}

hipError_t slab_partition(const Geometry &g, const DeviceArrays &d, int value_type, int S, int bits, int shift,
                          uint32_t *hist, void *scan_tmp, size_t scan_tmp_bytes, int32_t *col2, void *val2,
                          uint32_t *key2, hipStream_t s)
{
    const size_t hist_lds = (size_t)(SLAB_BLOCK / OMEGA) * S * HIST_PAD * 4;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(k_slab_hist), hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)hist_lds);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(k_slab_hist, dim3((g.p + HIST_TILES - 1) / HIST_TILES), dim3(SLAB_BLOCK), hist_lds, s, g, d.col, S, bits, shift,
                       hist);
    e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    e = rocprim::exclusive_scan(scan_tmp, scan_tmp_bytes, hist, hist, 0u, (size_t)S * g.p, rocprim::plus<uint32_t>(), s);
    if (e != hipSuccess)
        return e;
    const size_t vs = value_type == CSR5HIP_F64 ? 8 : 4;
    const size_t lds = slab_scatter_lds(g, S, vs);
    if (value_type == CSR5HIP_F64)
        hipLaunchKernelGGL(k_slab_scatter<double>, dim3(g.p), dim3(SLAB_BLOCK), lds, s, g, d.row_ptr, d.tile_ptr, d.col,
                           (const double *)d.val, S, bits, shift, hist, col2, (double *)val2, key2);
    else
        hipLaunchKernelGGL(k_slab_scatter<float>, dim3(g.p), dim3(SLAB_BLOCK), lds, s, g, d.row_ptr, d.tile_ptr, d.col,
                           (const float *)d.val, S, bits, shift, hist, col2, (float *)val2, key2);
    return hipGetLastError();
}

hipError_t slab_scan_tmp_bytes(size_t items, size_t *bytes)
{
    uint32_t *null_u = nullptr;
    return rocprim::exclusive_scan(nullptr, *bytes, null_u, null_u, 0u, items, rocprim::plus<uint32_t>(), nullptr);
}

hipError_t slab_select_tmp_bytes(int nnz, size_t *bytes)
{
    *bytes = ((size_t)seg_blocks(nnz) + 1) * 4; // chunk counts, then (in place) chunk offsets and the total
    return hipSuccess;
}

// d_count <- m' (number of segments); `tmp` (slab_select_tmp_bytes) keeps the chunk offsets for slab_segments
hipError_t slab_count_segments(int nnz, const uint32_t *key2, void *tmp, unsigned int *d_count, hipStream_t s)
{
    const int blocks = seg_blocks(nnz);
    hipLaunchKernelGGL(k_slab_count_segments, dim3(blocks), dim3(SLAB_BLOCK), 0, s, nnz, key2, (unsigned int *)tmp);
    hipLaunchKernelGGL(k_slab_scan_counts, dim3(1), dim3(1024), 0, s, blocks, (unsigned int *)tmp, d_count);
    return hipGetLastError();
}

// row_ptr'[0 .. m') <- segment starts (after slab_count_segments on the same keys and tmp)
hipError_t slab_segments(int nnz, const uint32_t *key2, const void *tmp, int32_t *row_ptr2, unsigned char *rowidx, hipStream_t s)
{
    hipLaunchKernelGGL(k_slab_emit_segments, dim3(seg_blocks(nnz)), dim3(SLAB_BLOCK), 0, s, nnz, key2,
                       (const unsigned int *)tmp, row_ptr2, rowidx);
    return hipGetLastError();
}

// tables of the combine: row byte per segment, run starts per (row block, slab), non-empty bit per parent row
size_t slab_base_words(int m, int S) { return ((size_t)(m + COMBINE_ROWS - 1) / COMBINE_ROWS + 2) * (size_t)S; }
hipError_t slab_tables(int m, int m2, int nnz, int S, int p, const int32_t *row_ptr, int32_t *row_ptr2, const uint32_t *key2,
                       const uint32_t *chunk_start, uint32_t *base, uint32_t *nonempty, hipStream_t s)
{
    const int nblk = (m + COMBINE_ROWS - 1) / COMBINE_ROWS;
    const long long entries = (long long)(nblk + 1) * S;
    hipLaunchKernelGGL(k_slab_base, dim3((unsigned)((entries + SLAB_BLOCK - 1) / SLAB_BLOCK)), dim3(SLAB_BLOCK), 0, s, m2, nnz, S,
                       nblk, row_ptr2, key2, chunk_start, p, base, row_ptr2 + m2);
    hipLaunchKernelGGL(k_slab_nonempty, dim3((m + 32 + SLAB_BLOCK - 1) / SLAB_BLOCK), dim3(SLAB_BLOCK), 0, s, m, row_ptr, nonempty);
    return hipGetLastError();
}

#if defined(CSR5_COMBINE_PERSISTENT) // experiment builds: one process-wide ticket word (the product would keep one per handle)
__device__ unsigned g_combine_ticket = 0;
#endif
template <typename VT>
static hipError_t combine_typed(int m, int tail_start, int zero_empty, int S, int m2, const uint32_t *base,
                                const unsigned char *rowidx, const uint32_t *nonempty, const void *P, void *y, hipStream_t s)
{
    const int rows_per_block = COMBINE_ROWS * COMBINE_WAVES;
    const dim3 grid((m + rows_per_block - 1) / rows_per_block), block(COMBINE_WAVES * OMEGA);
#if defined(CSR5_COMBINE_PERSISTENT)
    if (S == 16 && (int)grid.x > CSR5_COMBINE_PERSISTENT * 256) {
        unsigned *ticket = nullptr;
        hipError_t e = hipGetSymbolAddress((void **)&ticket, HIP_SYMBOL(g_combine_ticket));
        if (e != hipSuccess)
            return e;
        hipLaunchKernelGGL((k_slab_combine_persistent<VT, 16>), dim3(CSR5_COMBINE_PERSISTENT * 256), block, 0, s, m, tail_start,
                           zero_empty, m2, base, rowidx, nonempty, (const VT *)P, (VT *)y, ticket);
        return hipGetLastError();
    }
#endif
    switch (S) {
#define CSR5_SLAB_CASE(N)                                                                                              \
    case N:                                                                                                            \
        hipLaunchKernelGGL((k_slab_combine<VT, N>), grid, block, 0, s, m, tail_start, zero_empty, m2, base, rowidx,    \
                           nonempty, (const VT *)P, (VT *)y);                                                          \
        break;
        CSR5_SLAB_CASE(2) CSR5_SLAB_CASE(4) CSR5_SLAB_CASE(8) CSR5_SLAB_CASE(16) CSR5_SLAB_CASE(32) CSR5_SLAB_CASE(64)
#undef CSR5_SLAB_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_slab_combine(int m, int tail_start, int zero_empty, int S, int value_type, const uint32_t *base,
                               const unsigned char *rowidx, const uint32_t *nonempty, const void *P, int segments, void *y,
                               hipStream_t s)
{
    if (m <= 0 || segments <= 0)
        return hipSuccess;
    return value_type == CSR5HIP_F64 ? combine_typed<double>(m, tail_start, zero_empty, S, segments, base, rowidx, nonempty, P, y, s)
                                     : combine_typed<float>(m, tail_start, zero_empty, S, segments, base, rowidx, nonempty, P, y, s);
}

} // namespace csr5

namespace csr5 {

// Selects the hot columns of every slab and fills hot_cols / hot_count; *covered = sampled non-zeros whose column got a
// slot.  Needs only the column indices (any order: the parent's array), so it runs BEFORE the partition and its verdict
// can still change the slab count.  cnt, hotmap, chist, thr: caller-provided scratch (n words, slab_hotmap_bytes,
// S*HOT_BUCKETS, 2 S words; all zeroed by the caller except thr).
hipError_t slab_hot_select(int n, int nnz, int S, int bits, int shift, int capacity, int min_count, int sample_stride,
                           const int32_t *col, uint32_t *cnt, void *hotmap, uint32_t *chist, uint32_t *thr,
                           int32_t *hot_cols, int32_t *hot_count, unsigned long long *covered, hipStream_t s)
{
    const size_t G = slab_hot_groups(n, shift, bits);
    uint4 *hotbits = (uint4 *)hotmap;
    uint16_t *hotpre = (uint16_t *)(hotbits + (size_t)S * G);
    long long blocks = ((long long)nnz / sample_stride + COUNT_BLOCK * 8 - 1) / (COUNT_BLOCK * 8);
    blocks = blocks < 1 ? 1 : (blocks > COUNT_WGS ? COUNT_WGS : blocks);
    const size_t count_lds = (size_t)COUNT_SLOTS * 8;
    hipError_t ea = hipFuncSetAttribute(reinterpret_cast<const void *>(k_col_count), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)count_lds);
    if (ea != hipSuccess)
        return ea;
    // hot_count doubles as the fill counter of the second-class columns until k_hot_rank writes the slot counts
    // (slot 0 of every table is reserved -- it holds +0.0, see k_spmv_range -- so those start at 1)
    hipError_t e = hipMemsetAsync(hot_count, 0, (size_t)S * 4, s);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(k_col_count, dim3((unsigned)blocks), dim3(COUNT_BLOCK), count_lds, s, nnz, sample_stride, col, cnt);
    hipLaunchKernelGGL(k_hot_hist, dim3((n + HOT_HIST_COLS - 1) / HOT_HIST_COLS), dim3(SLAB_BLOCK), 0, s, n, cnt, S, bits, shift,
                       chist);
    hipLaunchKernelGGL(k_hot_threshold, dim3(S), dim3(OMEGA), 0, s, capacity, min_count, chist, thr);
    hipLaunchKernelGGL(k_hot_assign, dim3((n + HOT_ASSIGN_COLS - 1) / HOT_ASSIGN_COLS), dim3(SLAB_BLOCK), 0, s, n, S, cnt, bits, shift,
                       thr, min_count, hot_count, (uint32_t *)hotbits, G, covered);
    hipLaunchKernelGGL(k_hot_rank, dim3(S, (unsigned)((G + RANK_BLOCK - 1) / RANK_BLOCK)), dim3(RANK_BLOCK), 0, s, bits, shift,
                       capacity, G, hotbits, hotpre, hot_cols, hot_count);
    return hipGetLastError();
}

// after the partition: clamps the slot counts, first tile owned by every slab and the slab offsets
// (chunk_start = the scanned (slab, tile) histogram of slab_partition; p_hist = tiles of the parent, p / T = the child's)
hipError_t slab_hot_finish(int S, int p_hist, int p, int T, int nnz, int capacity, const uint32_t *chunk_start,
                           int32_t *hot_count, int32_t *tile0, int32_t *slab_off, hipStream_t s)
{
    hipLaunchKernelGGL(k_hot_finish, dim3(1), dim3(SLAB_MAX + 1), 0, s, S, p_hist, p, T, nnz, capacity, chunk_start,
                       hot_count, tile0, slab_off);
    return hipGetLastError();
}

// mode ENC_MARK: ref[slab][local id] = 1 for cold uses; ENC_PACK: 3-byte codes into col_lo / col_hi, cold columns coded by
// rank_of; col2 is only read
static hipError_t hot_encode_pass(int mode, int n, int nnz, int T, int p, int S, int bits, int shift, const int32_t *slab_off,
                                  const void *hotmap, int32_t *col2, uint16_t *col_lo, uint8_t *col_hi, uint8_t *ref,
                                  const uint32_t *rank_of, const uint32_t *key, hipStream_t s)
{
    const size_t G = slab_hot_groups(n, shift, bits);
    const size_t L = slab_local_count(n, shift, bits);
    const uint4 *hotbits = (const uint4 *)hotmap;
    const uint16_t *hotpre = (const uint16_t *)(hotbits + (size_t)S * G);
    const size_t lds = G * (sizeof(uint4) + sizeof(uint16_t));
    const dim3 grid(NUM_XCD * ENCODE_WGS_PER_XCD), block(ENCODE_BLOCK);
    const bool in_lds = lds <= 150 * 1024;
    auto launch = [&](auto kern) -> hipError_t {
        if (in_lds) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                               (int)lds);
            if (e != hipSuccess)
                return e;
        }
        hipLaunchKernelGGL(kern, grid, block, in_lds ? lds : 0, s, nnz, T, p, S, bits, shift, slab_off, hotbits, hotpre, G, col2,
                           col_lo, col_hi, ref, rank_of, L, key);
        return hipGetLastError();
    };
    if (mode == ENC_MARK)
        return in_lds ? launch(k_hot_encode<true, ENC_MARK>) : launch(k_hot_encode<false, ENC_MARK>);
    return in_lds ? launch(k_hot_encode<true, ENC_PACK>) : launch(k_hot_encode<false, ENC_PACK>);
}

// scratch of slab_hot_pack: marks (1 byte), ranks, sort keys in and out, sorted sources (S L entries each) + the sort's own
size_t slab_cold_words(int n, int S, int bits, int shift) { return (size_t)S * slab_local_count(n, shift, bits); }
hipError_t slab_cold_sort_tmp_bytes(size_t total, int slab_bits, size_t *bytes)
{
    uint32_t *nu = nullptr;
    rocprim::counting_iterator<uint32_t> iota(0u);
    return rocprim::radix_sort_pairs(nullptr, *bytes, nu, nu, iota, nu, total, 0, slab_bits + (int)COLD_KEY_BITS, nullptr);
}

// The packed column codes of a hot child: marks the columns that are gathered cold, ranks each slab's marked columns by
// their sampled use counts `cnt` (cold_base[S + 1] = start of every slab's cold region, cold_cols = the column behind
// every cold entry, in the order of the permuted copy of x) and writes the 3-byte codes.  words = slab_cold_words();
// ref: `words` bytes, zeroed by the caller; rank_of .. src_sorted: `words` uint32 each.
hipError_t slab_hot_pack(int n, int nnz, int T, int p, int S, int bits, int shift, const int32_t *slab_off, const void *hotmap,
                         const uint32_t *cnt, const uint32_t *key2, int32_t *col2, uint16_t *col_lo, uint8_t *col_hi, uint8_t *ref, uint32_t *rank_of,
                         uint32_t *keys, uint32_t *keys_sorted, uint32_t *src_sorted, void *sort_tmp, size_t sort_tmp_bytes,
                         int32_t *cold_base, int32_t *cold_cols, hipStream_t s)
{
    const size_t L = slab_local_count(n, shift, bits), total = (size_t)S * L;
    hipError_t e = hot_encode_pass(ENC_MARK, n, nnz, T, p, S, bits, shift, slab_off, hotmap, col2, nullptr, nullptr, ref, nullptr, nullptr, s);
    if (e != hipSuccess)
        return e;
    const unsigned blocks = (unsigned)((total + SLAB_BLOCK - 1) / SLAB_BLOCK);
    hipLaunchKernelGGL(k_cold_keys, dim3(blocks), dim3(SLAB_BLOCK), 0, s, total, L, n, bits, shift, ref, cnt, keys);
    rocprim::counting_iterator<uint32_t> iota(0u);
    e = rocprim::radix_sort_pairs(sort_tmp, sort_tmp_bytes, keys, keys_sorted, iota, src_sorted, total, 0, bits + (int)COLD_KEY_BITS, s);
    if (e != hipSuccess)
        return e;
    hipLaunchKernelGGL(k_cold_layout, dim3(1), dim3(SLAB_MAX), 0, s, S, L, keys_sorted, cold_base);
    hipLaunchKernelGGL(k_cold_rank, dim3(blocks), dim3(SLAB_BLOCK), 0, s, total, L, bits, shift, src_sorted, cold_base, rank_of, cold_cols);
    e = hipGetLastError();
    if (e != hipSuccess)
        return e;
    return hot_encode_pass(ENC_PACK, n, nnz, T, p, S, bits, shift, slab_off, hotmap, col2, col_lo, col_hi, nullptr, rank_of, key2, s);
}

size_t slab_local_columns(int n, int bits, int shift) { return slab_local_count(n, shift, bits); }

// slab-major bitmap of the hot columns (16 bytes per 128 slab-local columns) followed by the 2-byte group prefixes
size_t slab_hotmap_bytes(int n, int S, int bits, int shift)
{
    return (size_t)S * slab_hot_groups(n, shift, bits) * (sizeof(uint4) + sizeof(uint16_t)) + 256;
}

int slab_hot_buckets() { return HOT_BUCKETS; }

} // namespace csr5

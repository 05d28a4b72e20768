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
// own slab(s): the eight L2s hold eight DIFFERENT parts of x.  P = A' x holds one partial per segment and
//      y[r] = sum over the slabs k in mask[r] of P[ base[r / 64][k] + (number of rows < r in r's 64-block with bit k) ]
// is applied by k_slab_combine (mask: S bits per row; base: S words per 64 rows).  Deterministic: partials are added
// in slab order, no floating-point atomics.
//
// Build (all on the device, reading the parent's tile-ordered column_index / value through the transpose map):
//   k_slab_hist     workgroup / tile: slab histogram of the tile                       -> hist[slab][tile]
//   exclusive scan  (rocprim) over hist in slab-major order = start of every (slab, tile) chunk in A'
//   k_slab_scatter  workgroup / tile: stable rank of every element inside its slab (wave ballots + LDS chunk table),
//                   scatter of column, value and the 64-bit key (slab << 32 | row)
//   segment starts  = positions where the key changes: counted, then compacted into row_ptr' (rocprim::select)
//   k_slab_tables   thread / segment: mask bit (atomicOr) and the per-64-row base index
#include "csr5_internal.h"

#include <rocprim/device/device_scan.hpp>
#include <rocprim/device/device_select.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

namespace csr5 {

constexpr int SLAB_BLOCK = 256;
constexpr int SLAB_MAX = 64;

__device__ __forceinline__ uint32_t slab_of(uint32_t col, int shift, int bits)
{
    uint32_t v = col >> shift, out = 0;
    const uint32_t mask = (1u << bits) - 1u;
    while (v) {
        out ^= v & mask;
        v >>= bits;
    }
    return out;
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

__global__ void __launch_bounds__(SLAB_BLOCK)
k_slab_hist(Geometry g, const int32_t *__restrict__ col, int S, int bits, int shift, uint32_t *__restrict__ hist)
{
    __shared__ uint32_t sh[SLAB_MAX];
    const int t = blockIdx.x;
    const size_t base = (size_t)t * g.tile_elems;
    const int E = (int)((size_t)g.nnz - base < (size_t)g.tile_elems ? (size_t)g.nnz - base : (size_t)g.tile_elems);
    if (threadIdx.x < SLAB_MAX)
        sh[threadIdx.x] = 0;
    __syncthreads();
    for (int q = threadIdx.x; q < E; q += SLAB_BLOCK)
        atomicAdd(&sh[slab_of((uint32_t)col[base + q], shift, bits)], 1u);
    __syncthreads();
    if ((int)threadIdx.x < S)
        hist[(size_t)threadIdx.x * g.p + t] = sh[threadIdx.x];
}

template <typename VT>
__global__ void __launch_bounds__(SLAB_BLOCK)
k_slab_scatter(Geometry g, const int32_t *__restrict__ row_ptr, const uint32_t *__restrict__ tile_ptr,
               const int32_t *__restrict__ col, const VT *__restrict__ val, int S, int bits, int shift,
               const uint32_t *__restrict__ chunk_start, int32_t *__restrict__ col2, VT *__restrict__ val2,
               unsigned long long *__restrict__ key2)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int T = g.tile_elems;
    VT *sval = reinterpret_cast<VT *>(smem);
    int32_t *scol = reinterpret_cast<int32_t *>(smem + (size_t)T * sizeof(VT));
    uint32_t *off = reinterpret_cast<uint32_t *>(smem + (size_t)T * (sizeof(VT) + 4));
    unsigned char *skey = reinterpret_cast<unsigned char *>(off + (size_t)(T / OMEGA) * S);

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
    __syncthreads();
    const int nchunks = (E + OMEGA - 1) / OMEGA;
    // phase 1: per 64-element chunk, how many elements go to each slab
    for (int ch = wave; ch < nchunks; ch += SLAB_BLOCK / OMEGA) {
        const int c = ch * OMEGA + lane;
        const bool valid = c < E;
        const int k = valid ? skey[c] : -1;
        if (lane < S)
            off[ch * S + lane] = 0;
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int kk = __builtin_amdgcn_readlane(k, leader);
            const unsigned long long b = __ballot(valid && k == kk);
            if (lane == leader)
                off[ch * S + kk] = (uint32_t)__popcll(b);
            todo &= ~b;
        }
    }
    __syncthreads();
    // phase 2: exclusive scan over the chunks for every slab, on top of the global start of (slab, tile)
    if (tid < S) {
        uint32_t run = chunk_start[(size_t)tid * g.p + t];
        for (int ch = 0; ch < nchunks; ch++) {
            const uint32_t v = off[ch * S + tid];
            off[ch * S + tid] = run;
            run += v;
        }
    }
    __syncthreads();
    // phase 3: stable destination of every element + its row
    const int rs = (int)(tile_ptr[t] & ROW_MASK);
    const int re = (int)(tile_ptr[t + 1] & ROW_MASK);
    for (int ch = wave; ch < nchunks; ch += SLAB_BLOCK / OMEGA) {
        const int c = ch * OMEGA + lane;
        const bool valid = c < E;
        const int k = valid ? skey[c] : -1;
        int rank = 0;
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int leader = __builtin_ctzll(todo);
            const int kk = __builtin_amdgcn_readlane(k, leader);
            const unsigned long long b = __ballot(valid && k == kk);
            if (k == kk)
                rank = __popcll(b & ((1ull << lane) - 1ull));
            todo &= ~b;
        }
        if (valid) {
            const size_t dst = (size_t)off[ch * S + k] + rank;
            const int j = (int)(base + c);
            const int row = rs + upper_bound_i32(row_ptr + rs + 1, j, re - rs);
            col2[dst] = scol[c];
            val2[dst] = sval[c];
            key2[dst] = ((unsigned long long)k << 32) | (unsigned)row;
        }
    }
}

struct SegmentStart {
    const unsigned long long *key;
    __device__ bool operator()(const int &j) const { return j == 0 || key[j] != key[j - 1]; }
};

__global__ void __launch_bounds__(SLAB_BLOCK)
k_slab_count_segments(int nnz, const unsigned long long *__restrict__ key, unsigned int *__restrict__ count)
{
    unsigned local = 0;
    for (size_t j = (size_t)blockIdx.x * SLAB_BLOCK + threadIdx.x; j < (size_t)nnz; j += (size_t)gridDim.x * SLAB_BLOCK)
        local += j == 0 || key[j] != key[j - 1];
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
        local += __shfl_xor(local, d, OMEGA);
    if ((threadIdx.x & (OMEGA - 1)) == 0 && local)
        atomicAdd(count, local);
}

// one thread per segment s: mask bit of (row, slab); the first segment of a slab inside a 64-row block records its
// index as the block's base for that slab
__global__ void __launch_bounds__(SLAB_BLOCK)
k_slab_tables(int m2, int nnz, int S, const int32_t *__restrict__ row_ptr2, const unsigned long long *__restrict__ key,
              uint32_t *__restrict__ mask, uint32_t *__restrict__ base, int32_t *__restrict__ row_ptr2_end)
{
    const int s = blockIdx.x * SLAB_BLOCK + threadIdx.x;
    if (s == 0)
        *row_ptr2_end = nnz; // row_ptr'[m'] (the select wrote the m' starts)
    if (s >= m2)
        return;
    const unsigned long long kv = key[row_ptr2[s]];
    const uint32_t r = (uint32_t)kv, k = (uint32_t)(kv >> 32);
    const unsigned long long bit = (unsigned long long)r * S + k;
    atomicOr(&mask[bit >> 5], 1u << (bit & 31));
    bool first = s == 0;
    if (!first) {
        const unsigned long long pv = key[row_ptr2[s - 1]];
        first = (uint32_t)(pv >> 32) != k || ((uint32_t)pv >> 6) != (r >> 6);
    }
    if (first)
        base[(size_t)(r >> 6) * S + k] = (uint32_t)s;
}

// y[r] = sum of the partials of row r, in slab order.  One thread per row, one wavefront per 64-row block.
template <typename VT, int S>
__global__ void __launch_bounds__(SLAB_BLOCK)
k_slab_combine(int m, int tail_start, int zero_empty, const uint32_t *__restrict__ mask,
               const uint32_t *__restrict__ base, const VT *__restrict__ P, VT *__restrict__ y)
{
    const int r = blockIdx.x * SLAB_BLOCK + threadIdx.x;
    const int lane = threadIdx.x & (OMEGA - 1);
    const bool valid = r < m;
    unsigned long long mk = 0;
    if (valid) {
        if constexpr (S == 64) {
            mk = (unsigned long long)mask[2 * (size_t)r] | ((unsigned long long)mask[2 * (size_t)r + 1] << 32);
        } else {
            const unsigned long long bit = (unsigned long long)r * S;
            const uint32_t w = mask[bit >> 5] >> (bit & 31);
            mk = S == 32 ? w : (w & ((1u << (S & 31)) - 1u));
        }
    }
    if (!__ballot(mk != 0)) {
        if (valid && (zero_empty || r >= tail_start))
            y[r] = 0;
        return;
    }
    const uint32_t bv = lane < S ? base[(size_t)(r >> 6) * S + lane] : 0u;
    const unsigned long long lt = (1ull << lane) - 1ull;
    VT sum = 0;
    constexpr int CH = S < 16 ? S : 16; // loads in flight per lane
#pragma unroll
    for (int k0 = 0; k0 < S; k0 += CH) {
        VT part[CH];
#pragma unroll
        for (int j = 0; j < CH; j++) {
            const int k = k0 + j;
            const bool bit = (mk >> k) & 1ull;
            const unsigned long long b = __ballot(bit);
            const uint32_t idx = (uint32_t)__builtin_amdgcn_readlane((int)bv, k) + (uint32_t)__popcll(b & lt);
            part[j] = bit ? P[idx] : (VT)0;
        }
#pragma unroll
        for (int j = 0; j < CH; j++)
            sum += part[j];
    }
    if (!valid)
        return;
    if (mk)
        y[r] = sum;
    else if (zero_empty || r >= tail_start)
        y[r] = 0;
}

// ---- host side -----------------------------------------------------------------------------------------
size_t slab_scatter_lds(const Geometry &g, int S, size_t vsize)
{
    const size_t T = (size_t)g.tile_elems;
    return T * (vsize + 4) + (T / OMEGA) * (size_t)S * 4 + T;
}

hipError_t slab_partition(const Geometry &g, const DeviceArrays &d, int value_type, int S, int bits, int shift,
                          uint32_t *hist, void *scan_tmp, size_t scan_tmp_bytes, int32_t *col2, void *val2,
                          unsigned long long *key2, hipStream_t s)
{
    hipLaunchKernelGGL(k_slab_hist, dim3(g.p), dim3(SLAB_BLOCK), 0, s, g, d.col, S, bits, shift, hist);
    hipError_t e = hipGetLastError();
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

hipError_t slab_count_segments(int nnz, const unsigned long long *key2, unsigned int *d_count, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(d_count, 0, 4, s);
    if (e != hipSuccess)
        return e;
    long long blocks = ((long long)nnz + SLAB_BLOCK * 16 - 1) / (SLAB_BLOCK * 16);
    blocks = blocks < 1 ? 1 : (blocks > 65536 ? 65536 : blocks);
    hipLaunchKernelGGL(k_slab_count_segments, dim3((unsigned)blocks), dim3(SLAB_BLOCK), 0, s, nnz, key2, d_count);
    return hipGetLastError();
}

hipError_t slab_select_tmp_bytes(int nnz, size_t *bytes)
{
    int32_t *null_i = nullptr;
    unsigned int *null_c = nullptr;
    return rocprim::select(nullptr, *bytes, rocprim::counting_iterator<int>(0), null_i, null_c, (size_t)nnz,
                           SegmentStart{nullptr}, nullptr);
}

hipError_t slab_segments(int nnz, const unsigned long long *key2, void *tmp, size_t tmp_bytes, int32_t *row_ptr2,
                         unsigned int *d_count, hipStream_t s)
{
    return rocprim::select(tmp, tmp_bytes, rocprim::counting_iterator<int>(0), row_ptr2, d_count, (size_t)nnz,
                           SegmentStart{key2}, s);
}

hipError_t slab_tables(int m2, int nnz, int S, int32_t *row_ptr2, const unsigned long long *key2, uint32_t *mask,
                       uint32_t *base, hipStream_t s)
{
    const int blocks = ((m2 > 0 ? m2 : 1) + SLAB_BLOCK - 1) / SLAB_BLOCK;
    hipLaunchKernelGGL(k_slab_tables, dim3(blocks), dim3(SLAB_BLOCK), 0, s, m2, nnz, S, row_ptr2, key2, mask, base,
                       row_ptr2 + m2);
    return hipGetLastError();
}

template <typename VT>
static hipError_t combine_typed(int m, int tail_start, int zero_empty, int S, const uint32_t *mask, const uint32_t *base,
                                const void *P, void *y, hipStream_t s)
{
    const dim3 grid((m + SLAB_BLOCK - 1) / SLAB_BLOCK), block(SLAB_BLOCK);
    switch (S) {
#define CSR5_SLAB_CASE(N)                                                                                              \
    case N:                                                                                                            \
        hipLaunchKernelGGL((k_slab_combine<VT, N>), grid, block, 0, s, m, tail_start, zero_empty, mask, base,          \
                           (const VT *)P, (VT *)y);                                                                    \
        break;
        CSR5_SLAB_CASE(2) CSR5_SLAB_CASE(4) CSR5_SLAB_CASE(8) CSR5_SLAB_CASE(16) CSR5_SLAB_CASE(32) CSR5_SLAB_CASE(64)
#undef CSR5_SLAB_CASE
    default: return hipErrorInvalidValue;
    }
    return hipGetLastError();
}

hipError_t launch_slab_combine(int m, int tail_start, int zero_empty, int S, int value_type, const uint32_t *mask,
                               const uint32_t *base, const void *P, void *y, hipStream_t s)
{
    if (m <= 0)
        return hipSuccess;
    return value_type == CSR5HIP_F64 ? combine_typed<double>(m, tail_start, zero_empty, S, mask, base, P, y, s)
                                     : combine_typed<float>(m, tail_start, zero_empty, S, mask, base, P, y, s);
}

} // namespace csr5

// csr5_slabmap.h -- the column -> (slab, slab-local id) map of the column-slab structure (csr5_slab.hip builds with it,
// csr5_hot.hip decodes packed column codes with it).  Device code only.
#pragma once
#include <hip/hip_runtime.h>
#include <cstddef>
#include <cstdint>

namespace csr5 {

// slab of a column = xor of all `bits`-wide groups of (col >> shift).  (A fold by halves -- five uniform steps instead of a
// loop whose length depends on the column -- was slower: most columns need two or three rounds.)
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

// Columns of one slab, renumbered densely: the slab is the xor of all `bits`-wide groups of (col >> shift), so given the slab
// the LOWEST group is determined by the others -- dropping it is a bijection from the slab's columns onto [0, L),
// L = slab_local_count(n).  The hot map is stored slab after slab in these local ids, as a BITMAP (bit = the column owns a
// table slot) plus the number of set bits in front of every 128-bit group: slots are handed out in column order, so the
// slot of a hot column is that prefix + the set bits below it in its group.  One slab's slice is 18 bytes per 128 columns
// (144 KB for R-MAT 24 at 16 slabs): k_hot_encode keeps it in LDS.  (A map of 2 bytes per column cost one L1 line fill
// per non-zero: 1.7 ms on R-MAT 24.)
__host__ __device__ inline uint32_t slab_local(uint32_t col, int shift, int bits)
{
    return ((col >> (shift + bits)) << shift) | (col & ((1u << shift) - 1u));
}
__host__ __device__ inline size_t slab_local_count(int n, int shift, int bits)
{
    return ((size_t)(((uint32_t)(n > 0 ? n - 1 : 0)) >> (shift + bits)) + 1) << shift;
}
// the column of slab k with local id `local` (inverse of slab_local on that slab)
__device__ __forceinline__ uint32_t slab_column(uint32_t k, uint32_t local, int shift, int bits)
{
    const uint32_t hi = local >> shift, lo = local & ((1u << shift) - 1u);
    const uint32_t low_group = k ^ slab_of(hi << shift, shift, bits); // xor of the groups above the lowest one
    return (hi << (shift + bits)) | (low_group << shift) | lo;
}
// The same without a data-dependent loop, for code whose LATENCY matters (k_spmv_range decodes a tile's column codes in
// front of its gathers): the groups above the lowest one are folded by halves with wave-uniform shift counts; a count
// beyond the id's width shifts everything out (local ids stay below 2^23).
__device__ __forceinline__ uint32_t slab_column_straight(uint32_t k, uint32_t local, int shift, int bits)
{
    const uint32_t hi = local >> shift;
    uint32_t v = hi;
#pragma unroll
    for (int step = 16; step >= 1; step >>= 1) {
        const int s = step * bits;
        v ^= v >> (s < 31 ? s : 31);
    }
    const uint32_t low_group = (k ^ v) & ((1u << bits) - 1u);
    return (hi << (shift + bits)) | (low_group << shift) | (local & ((1u << shift) - 1u));
}
} // namespace csr5

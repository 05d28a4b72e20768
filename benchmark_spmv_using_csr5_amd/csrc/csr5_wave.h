// csr5_wave.h -- wave64 cross-lane helpers shared by the SpMV kernels (csr5_spmv.hip, csr5_hot.hip); gfx950 only.
#pragma once

#include "csr5_internal.h"

namespace csr5 {

// a * b + c as ONE fused multiply-add of the value type (__builtin_fma alone is the fp64 one: fp32 operands would be widened,
// multiplied-added in fp64 and narrowed again -- three conversions per element and the fp64 rate; so it was until round 5)
__device__ __forceinline__ float fma_vt(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
__device__ __forceinline__ double fma_vt(double a, double b, double c) { return __builtin_fma(a, b, c); }

// ---- cross-lane helpers on DPP (data-parallel primitives: lane moves folded into VALU operands, no LDS
//      crossbar round trip as with ds_bpermute).  A 64-bit value moves as two 32-bit halves. -------------
// Full row/bank masks: bound_ctrl makes source lanes outside the row / wave read 0 and leaves no "old"
// operand to initialise (2 VALU less per 64-bit move).  Partial masks: masked lanes keep old = 0.
template <int CTRL, int ROW_MASK_ = 0xF, int BANK_MASK_ = 0xF>
__device__ __forceinline__ int dpp_word(int w)
{
    constexpr bool FULL = ROW_MASK_ == 0xF && BANK_MASK_ == 0xF;
    return __builtin_amdgcn_update_dpp(0, w, CTRL, ROW_MASK_, BANK_MASK_, FULL);
}
template <int CTRL, int ROW_MASK_ = 0xF, int BANK_MASK_ = 0xF>
__device__ __forceinline__ float dpp_move(float v)
{
    return __builtin_bit_cast(float, dpp_word<CTRL, ROW_MASK_, BANK_MASK_>(__builtin_bit_cast(int, v)));
}
template <int CTRL, int ROW_MASK_ = 0xF, int BANK_MASK_ = 0xF>
__device__ __forceinline__ double dpp_move(double v)
{
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const int lo = dpp_word<CTRL, ROW_MASK_, BANK_MASK_>((int)(unsigned)b);
    const int hi = dpp_word<CTRL, ROW_MASK_, BANK_MASK_>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo);
}
// lanes masked off by row/bank masks or shifted in from outside a row read 0 (old = 0, bound_ctrl off)
constexpr int DPP_ROW_SHR1 = 0x111, DPP_ROW_SHR2 = 0x112, DPP_ROW_SHR4 = 0x114, DPP_ROW_SHR8 = 0x118;
constexpr int DPP_ROW_SHL1 = 0x101, DPP_ROW_SHL2 = 0x102, DPP_ROW_SHL4 = 0x104, DPP_ROW_SHL8 = 0x108;
constexpr int DPP_ROW_BCAST15 = 0x142, DPP_ROW_BCAST31 = 0x143, DPP_WAVE_SHL1 = 0x130;

// value of lane `src` (wave-uniform index) in every lane: v_readlane, no LDS crossbar trip
template <typename VT>
__device__ __forceinline__ VT bcast_lane(VT v, int src)
{
    if constexpr (sizeof(VT) == 8) {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)b, src);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), src);
        return __builtin_bit_cast(VT, ((unsigned long long)hi << 32) | lo);
    } else {
        return __builtin_bit_cast(VT, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), src));
    }
}

// sum over the 64 lanes, result in every lane (6 DPP steps + one readlane broadcast).  Only lane 63 has
// to end up right, so the two row broadcasts run with full row masks as well: rows that receive a value
// they should not are never read again.
template <typename VT>
__device__ __forceinline__ VT wave_sum(VT v)
{
    v += dpp_move<DPP_ROW_SHR1>(v);                 // pairs
    v += dpp_move<DPP_ROW_SHR2>(v);                 // quads
    v += dpp_move<DPP_ROW_SHR4>(v);                 // 8
    v += dpp_move<DPP_ROW_SHR8>(v);                 // lane 15 of every row holds the row sum
    v += dpp_move<DPP_ROW_BCAST15>(v);              // lane 16r+15 += total of row r-1
    v += dpp_move<DPP_ROW_BCAST31>(v);              // rows 2,3 += lane 31 -> lane 63 = wave sum
    return bcast_lane(v, OMEGA - 1);
}
// bitwise OR over the 64 lanes, result wave-uniform (same DPP steps as wave_sum)
__device__ __forceinline__ uint32_t wave_or(uint32_t w)
{
    int v = (int)w;
    v |= dpp_word<DPP_ROW_SHR1>(v);
    v |= dpp_word<DPP_ROW_SHR2>(v);
    v |= dpp_word<DPP_ROW_SHR4>(v);
    v |= dpp_word<DPP_ROW_SHR8>(v);
    v |= dpp_word<DPP_ROW_BCAST15>(v);
    v |= dpp_word<DPP_ROW_BCAST31>(v);
    return (uint32_t)__builtin_amdgcn_readlane(v, OMEGA - 1);
}
// sum over lanes 0..count-1 of a value that is ZERO in every other lane (count wave-uniform, 1..64)
template <typename VT>
__device__ __forceinline__ VT head_sum(VT v, int count)
{
    if (count <= 4) {
        v += dpp_move<DPP_ROW_SHR1>(v);
        v += dpp_move<DPP_ROW_SHR2>(v);
        return bcast_lane(v, 3);
    }
    if (count <= 16) {
        v += dpp_move<DPP_ROW_SHR1>(v);
        v += dpp_move<DPP_ROW_SHR2>(v);
        v += dpp_move<DPP_ROW_SHR4>(v);
        v += dpp_move<DPP_ROW_SHR8>(v);
        return bcast_lane(v, 15);
    }
    return wave_sum(v);
}
// inclusive prefix sum over the lanes (lane l gets v[0] + ... + v[l]): six DPP adds -- four Hillis-Steele steps inside the
// rows of 16 (row_shr fills with zeros), then lane 15 of rows 0 / 2 into rows 1 / 3 and lane 31 into rows 2 and 3
__device__ __forceinline__ int wave_scan_incl(int v)
{
    v += dpp_word<DPP_ROW_SHR1>(v);
    v += dpp_word<DPP_ROW_SHR2>(v);
    v += dpp_word<DPP_ROW_SHR4>(v);
    v += dpp_word<DPP_ROW_SHR8>(v);
    v += dpp_word<DPP_ROW_BCAST15, 0xA>(v);
    v += dpp_word<DPP_ROW_BCAST31, 0xC>(v);
    return v;
}
// value of lane l+1 (lane 63 receives 0)
template <typename VT>
__device__ __forceinline__ VT lane_above(VT v)
{
    return dpp_move<DPP_WAVE_SHL1>(v);
}

} // namespace csr5

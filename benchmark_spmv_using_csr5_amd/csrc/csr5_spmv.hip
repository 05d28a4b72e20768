// csr5_spmv.hip -- CSR5 SpMV for gfx950 (wave64), written from scratch.
//
// Reference decomposition (CSR5_cuda/detail/cuda/csr5_spmv_cuda.h):
//   K9  spmv_csr5_compute_kernel        :275-311   warp / tile, calibrator[tile]
//   K10 spmv_csr5_calibrate_kernel      :313-382   y[tile_ptr[t]] += calibrator[t], atomics at block edges
//   K11 spmv_csr5_tail_partition_kernel :384-419   one 32-thread block per tail ROW
// i.e. three dependent launches per SpMV and a y that the caller must have zeroed.
//
// Here (one wavefront = one tile, omega = 64, one tile per 64-thread workgroup):
//   fused mode    : k_spmv<.., FUSED=true>  ONE launch per SpMV [default].  Tiles, the CSR tail (extra
//                   workgroups of the same grid) and the carries.  A row that starts in tile t and spills
//                   <= 64 elements into tile t+1 (the common case) is finished by tile t itself, which
//                   re-reads those few elements: nothing is communicated.  Other cut rows resolve at the
//                   slot of the first tile of the row's run: 1 partial -> plain store, 2 -> one-atomic
//                   exchange handshake, > 2 -> per-party words + arrival counter, summed in tile order by the
//                   last arriver.  No spinning, bit-reproducible, y need not be zeroed.
//   two-pass mode : k_spmv<.., FUSED=false> + k_calibrate: carries added in tile order by a second launch
//                   (the reference's compute -> calibrate structure).
// Variants chosen at conversion: XWIN (a 4-KB slice of x per tile staged in LDS, in-window lanes gather
// with ds_read), LDSY (a tile's y segments compacted in LDS, flushed with coalesced stores) and NT (non-temporal
// column/value streams for matrices beyond the Infinity Cache).
// Lane-local work walks the bit flags held in ONE 32-bit register (sigma <= 32); the cross-lane step is a
// flag-propagating backward segmented scan over the 64 lanes (no prefix-sum difference, so no
// cancellation) on DPP row shifts and v_readlane row carries; every independent load of a tile is in flight before any loaded
// value is consumed (two memory round trips per tile); column_index/value loads are fully coalesced
// 256-B / 512-B wave accesses thanks to the tile transpose.
#include "csr5_carry.h"

namespace csr5 {

// ---- per-wavefront LDS region of the tile kernel ---------------------------------------------------
// LDSY: the y segments of a tile (<= T = 64*sigma values) are first written to LDS at their segment index
// and then flushed with coalesced stores (one 512-B wave store per 64 rows instead of up to sigma
// scattered, partially masked 8-byte stores).  Used while T * sizeof(vT) <= 8 KiB per wavefront
// (fp64: sigma <= 16, fp32: sigma <= 32) so occupancy is not LDS-limited.
// The x-window (XWIN) shares the region: the window is dead once the gathers have returned.
template <typename VT, int SIGMA, bool LDSY_REQ>
constexpr bool use_ldsy() { return LDSY_REQ && SIGMA > 0 && (size_t)OMEGA * SIGMA * sizeof(VT) <= 8192; }
template <typename VT, int SIGMA, bool XWIN, bool LDSY_REQ>
constexpr int wave_lds_bytes()
{
    int b = use_ldsy<VT, SIGMA, LDSY_REQ>() ? OMEGA * SIGMA * (int)sizeof(VT) : 0;
    if (XWIN && b < XWIN_BYTES)
        b = XWIN_BYTES;
    return (b + 15) & ~15;
}

#if defined(CSR5_TILE_STAMPS) // experiment builds only: wall-clock stamps (100 MHz) per tile: start, all loads + gathers landed, end
static __device__ unsigned long long g_tile_stamps[3 * (1 << 16)];
#if defined(CSR5_SPMV_ONLY_F32) // (the file is compiled once per value type: one array and one export per half)
extern "C" int csr5hip_debug_tile_stamps_f32(unsigned long long *dst, int count)
#else
extern "C" int csr5hip_debug_tile_stamps_f64(unsigned long long *dst, int count)
#endif
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_tile_stamps), (size_t)count * sizeof(unsigned long long));
}
#define TILE_STAMP(i, wait)                                                                                            \
    do {                                                                                                               \
        if (wait)                                                                                                      \
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                                                           \
        if (lane == 0 && t < (1 << 16))                                                                                \
            g_tile_stamps[3 * t + (i)] = wall_clock64();                                                               \
    } while (0)
#else
#define TILE_STAMP(i, wait)
#endif

// ---- tiles 0..p-2 ------------------------------------------------------------------------------
// SIGMA > 0: compile-time sigma (loads hoisted into registers, flag walk fully unrolled).
// SIGMA == 0: run-time sigma (any 1..32), same code shape, used for sigma < 4 and as a cross-check.
// One tile, one wavefront.  `wave_lds` = this wavefront's private LDS region (x-window / y segments).
template <typename VT, int SIGMA, bool FUSED, bool XWIN, bool LDSY_REQ, bool NT, bool C16 = false, bool C31 = false>
__device__ __forceinline__ void
tile_body(const Geometry &g, const int t, const int lane, const int32_t *__restrict__ col, const VT *__restrict__ val,
          const VT *__restrict__ x, const uint32_t *__restrict__ tile_ptr, const uint32_t *__restrict__ tile_desc,
          const int32_t *__restrict__ offset_ptr, const int32_t *__restrict__ offset, VT *__restrict__ calibrator,
          VT *__restrict__ y, VT *acc, uint32_t *cnt, const uint4 *__restrict__ meta, const uint32_t *__restrict__ hdr,
          char *wave_lds, const uint32_t *__restrict__ col16 = nullptr, const int32_t *__restrict__ base16 = nullptr,
          const uint32_t *__restrict__ col31 = nullptr)
{
    static_assert(!C16 || (SIGMA > 0 && SIGMA % 2 == 0 && FUSED && !NT), "narrow column codes: two per word, fused kernel");
    static_assert(!C31 || (SIGMA > 0 && FUSED && !XWIN && !C16), "flagged column words: plain fused kernel, compile-time sigma");
    auto gather = [&](int32_t cw) -> VT { return x[(uint32_t)cw]; };
    const int sigma = SIGMA > 0 ? SIGMA : g.sigma;
    const int bit_y = SIGMA > 0 ? bit_y_of(SIGMA > 0 ? SIGMA : 1) : g.bit_y;
    const int bit_all = bit_y + BIT_SS;
    const int num_packet = SIGMA > 0 ? num_packet_of(SIGMA > 0 ? SIGMA : 1) : g.num_packet;
    const int T = OMEGA * sigma;

    // ---- issue every independent load of the tile first: tile_ptr pair, carry meta, descriptor
    //      words, the short-spill elements of tile t+1 and the sigma column/value pairs.  No wait
    //      and no data-dependent branch sits between them, so one memory round trip covers all
    //      of them and a second one covers the x gathers.
    const size_t base = (size_t)t * T + lane;
    // (C31: the same words with the element's row-start flag in bit 31, from the kernel-side copy)
    const int32_t *ct = (C31 ? reinterpret_cast<const int32_t *>(col31) : col) + base;
    const VT *vt = val + base;
    const uint32_t *d = tile_desc + (size_t)t * OMEGA * num_packet;
    // The wave-uniform words (tile_ptr pair, carry meta) deliberately go through the VECTOR memory
    // path: vector loads return in order, so they share the tile loads' round trip, whereas scalar
    // loads would force an s_waitcnt lgkmcnt(0) (they return out of order) in front of the tile loads.
    // `vz` is an opaque per-lane zero that keeps the compiler from scalarising these loads.
    int vz;
    asm volatile("v_mov_b32 %0, 0" : "=v"(vz));
    uint32_t tp0 = 0, tp1 = 0, hw = 0;
    if constexpr (FUSED) {
        // fused mode: the whole 32-byte tile header (k_tile_tables) in one load, one word per lane (lanes 0..7)
        hw = hdr[8 * (size_t)t + (lane & 7)];
    } else {
        tp0 = tile_ptr[t + vz];
        tp1 = tile_ptr[t + 1 + vz];
    }
    uint4 mt = make_uint4(0u, 0u, 0u, 0u);
    uint32_t mt_next_x = 0;
    int32_t spill_c = 0;
    VT spill_v = 0;
    // Issue order matters: vector loads return in order and this first round trip is bandwidth-bound
    // (every resident wave streams its tile at once), so the words the SECOND round trip depends on --
    // the column indices -- are requested first; the x gathers then go out while the values (2/3 of the
    // bytes) are still streaming in, instead of after them.
    size_t spill_pos = 0;
    if constexpr (FUSED) {
        // first 64 elements (CSR order) of tile t+1: a transposed tile keeps element j at
        // (j % sigma)*omega + j / sigma, the CSR tail keeps it at j
        const size_t nb = (size_t)(t + 1) * T;
        size_t pos = (t + 1 == g.p - 1) ? nb + lane : nb + (size_t)(lane % sigma) * OMEGA + lane / sigma;
        spill_pos = pos < (size_t)g.nnz ? pos : (size_t)g.nnz - 1;
        if (!g.defer) // (deferred carries: nobody finishes a neighbour's spill -- see derive_kernel_tables)
            spill_c = col[spill_pos];
    }
    constexpr int NREG = SIGMA > 0 ? SIGMA : 1;
    int32_t c[NREG];
    VT v[NREG];
    int32_t base_c16 = 0; // (narrow column codes: word d of the lane waits in c[d] until it is decoded in place)
    if constexpr (SIGMA > 0) {
        if constexpr (NT) {
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                c[i] = __builtin_nontemporal_load(ct + i * OMEGA);
        } else if constexpr (C16) {
            // narrow column codes (k_col16): two per word, half the column stream, in 16-byte pieces per lane (col16_word_offset:
            // sigma / 8 loads per tile instead of sigma / 2 -- the address unit is busy 76 % of this kernel's launch and handles
            // one wave instruction per ~16 clocks whatever its width); the tile's base rides in the same batch
            const uint32_t *ctile = col16 + (size_t)t * (T / 2);
            constexpr int W = SIGMA / 2, G4 = W / 4;
#pragma unroll
            for (int k = 0; k < G4; k++) {
                const uint4 q = reinterpret_cast<const uint4 *>(ctile)[k * OMEGA + lane];
                c[4 * k] = (int32_t)q.x, c[4 * k + 1] = (int32_t)q.y, c[4 * k + 2] = (int32_t)q.z, c[4 * k + 3] = (int32_t)q.w;
            }
            if constexpr (W % 4 != 0) {
                const uint2 q = reinterpret_cast<const uint2 *>(ctile + G4 * 4 * OMEGA)[lane];
                c[4 * G4] = (int32_t)q.x, c[4 * G4 + 1] = (int32_t)q.y;
            }
            base_c16 = base16[t + vz];
        } else {
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                c[i] = ct[i * OMEGA];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    // (narrow column codes carry the bit flags themselves and y_offset is recomputed from them: no descriptor load)
    uint32_t w0 = 0, w1 = 0;
    if constexpr (!C16 && !C31) {
        w0 = d[lane];
        w1 = num_packet > 1 ? d[OMEGA + lane] : 0u;
    }
    uint32_t flags16 = 0;
    if constexpr (FUSED) {
        if (!g.defer)
            spill_v = val[spill_pos];
        if constexpr (SIGMA > 0)
            __builtin_amdgcn_sched_barrier(0); // keep it AHEAD of the value stream (see the pin below)
    }

    // this lane's sigma elements (coalesced: lane stride 1 at every step)
    VT mv[NREG], mx[NREG]; // matrix value and gathered x of element i: multiplied inside the fused multiply-adds below
    VT lead_next = 0;
    if constexpr (SIGMA > 0) {
        // a non-temporal hint on the streams helps only when the matrix is far larger than the 256-MiB
        // Infinity Cache (R-MAT 22: +5 %) and costs 18-25 % when it is not (R-MAT 20, nd24k-like), because
        // repeated SpMVs then re-stream from HBM: chosen per matrix by the host (CSR5HIP_OPT_STREAM_NT),
        // compiled as its own kernel variant (a run-time branch gets merged and loses the hint)
        if constexpr (NT) {
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                v[i] = __builtin_nontemporal_load(vt + i * OMEGA);
        } else {
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                v[i] = vt[i * OMEGA];
        }
        // everything above is in flight before anything below consumes a loaded value
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (FUSED) {
            mt = make_uint4(__builtin_amdgcn_readlane(hw, 0), __builtin_amdgcn_readlane(hw, 1),
                            __builtin_amdgcn_readlane(hw, 2), __builtin_amdgcn_readlane(hw, 3));
            mt_next_x = __builtin_amdgcn_readlane(hw, 4);
            tp0 = __builtin_amdgcn_readlane(hw, 5);
            tp1 = __builtin_amdgcn_readlane(hw, 6);
            // Pin the two spill loads to the first round trip: they are consumed only under `L > 0`, and the
            // optimiser otherwise sinks them below that test (behind the header's arrival: a third dependent round
            // trip for every tile with a short spill).  Costs nothing: vector loads return in order and these two
            // were requested right after the column words the gathers below wait for anyway.
            asm volatile("" : "+v"(spill_c), "+v"(spill_v));
        }
        if constexpr (C16) {
            const int32_t base = __builtin_amdgcn_readfirstlane(base_c16);
#pragma unroll
            for (int dd = 0; dd < SIGMA / 2; dd++) // the row-start flags (bits 15 and 31 of a word) first, into one register
                flags16 |= (((uint32_t)c[dd] >> 15) & 1u) << (31 - 2 * dd) | ((uint32_t)c[dd] >> 31) << (30 - 2 * dd);
#pragma unroll
            for (int dd = SIGMA / 2 - 1; dd >= 0; dd--) { // (downwards: word dd sits in c[dd], below the slots it decodes into)
                const uint32_t w = (uint32_t)c[dd];
                c[2 * dd + 1] = base + (int32_t)((w >> 16) & 0x7FFFu);
                c[2 * dd] = base + (int32_t)(w & 0x7FFFu);
            }
        }
        if constexpr (C31) {
#pragma unroll
            for (int i = 0; i < SIGMA; i++) { // element i's flag -> bit 31 - i (the descriptor's order), then the bare column
                flags16 |= ((uint32_t)c[i] >> 31) << (31 - i);
                c[i] &= 0x7FFFFFFF;
            }
        }
        VT xv[NREG];
        if constexpr (XWIN) {
            constexpr int XWIN_ELEMS = xwin_elems(sizeof(VT));
            // LDS x-window: carry_meta[t].w - 1 = first column of a XWIN_ELEMS-wide slice of x that
            // covers most of this tile's columns (chosen at conversion, k_tile_tables).  In-window lanes
            // gather from LDS (a ds_read costs a few cycles; a divergent global gather >= 34 clk per
            // wave instruction even on L1 hits); the others gather from memory as before and are
            // issued FIRST, so they overlap the window fetch.
            VT *win = reinterpret_cast<VT *>(wave_lds);
            const int wlo = (int)__builtin_amdgcn_readfirstlane(mt.w) - 1;
            const bool x_vec16 = (reinterpret_cast<uintptr_t>(x) & 15u) == 0; // (a caller's x need only be element-aligned)
            if (wlo >= 0) {
                // stage the window (private to this wave): four 16-byte loads and LDS stores per lane when x allows it (window
                // bases are multiples of 4 columns, k_tile_tables), else 16 coalesced element loads
                if (x_vec16 && wlo + XWIN_ELEMS <= g.n) {
                    const uint4 *src = reinterpret_cast<const uint4 *>(x + wlo);
                    uint4 *dst = reinterpret_cast<uint4 *>(win);
                    uint4 q[XWIN_BYTES / 16 / OMEGA];
#pragma unroll
                    for (int k = 0; k < XWIN_BYTES / 16 / OMEGA; k++)
                        q[k] = src[k * OMEGA + lane];
#pragma unroll
                    for (int k = 0; k < XWIN_BYTES / 16 / OMEGA; k++)
                        dst[k * OMEGA + lane] = q[k];
                } else {
#pragma unroll
                    for (int k = 0; k < XWIN_ELEMS / OMEGA; k++) {
                        const int j = wlo + k * OMEGA + lane;
                        win[k * OMEGA + lane] = x[j < g.n ? j : g.n - 1];
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
#pragma unroll
                for (int i = 0; i < SIGMA; i++) {
                    const unsigned dlt = (unsigned)(c[i] - wlo);
                    xv[i] = dlt < (unsigned)XWIN_ELEMS ? win[dlt] : x[(uint32_t)c[i]];
                }
            } else {
#pragma unroll
                for (int i = 0; i < SIGMA; i++)
                    xv[i] = x[(uint32_t)c[i]];
            }
        } else {
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                xv[i] = gather(c[i]);
        }
        if constexpr (FUSED) {
            // the closing row of this tile spills mt.z <= 64 elements into tile t+1 and ends there:
            // gather x for exactly those lanes; the other lanes re-read x[0] (one cache line), so the
            // gather is unconditional and rides in the same round trip as the tile's own gathers
            const int L = ((mt.x >> 29) & 1u) ? (int)mt.z : 0;
            if (!g.defer) {
                const VT sx = gather(lane < L ? spill_c : 0);
                lead_next = lane < L ? spill_v * sx : (VT)0;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < SIGMA; i++) {
            mv[i] = v[i];
            mx[i] = xv[i];
        }
        TILE_STAMP(1, true);
    } else if constexpr (FUSED) {
        mt = make_uint4(__builtin_amdgcn_readlane(hw, 0), __builtin_amdgcn_readlane(hw, 1),
                        __builtin_amdgcn_readlane(hw, 2), __builtin_amdgcn_readlane(hw, 3));
        mt_next_x = __builtin_amdgcn_readlane(hw, 4);
        tp0 = __builtin_amdgcn_readlane(hw, 5);
        tp1 = __builtin_amdgcn_readlane(hw, 6);
        const int L = ((mt.x >> 29) & 1u) ? (int)mt.z : 0;
        const VT sx = gather(lane < L ? spill_c : 0);
        lead_next = lane < L ? spill_v * sx : (VT)0;
    }
    auto product = [&](int i) -> VT {
        if constexpr (SIGMA > 0)
            return mv[i] * mx[i];
        else
            return vt[i * OMEGA] * gather(ct[i * OMEGA]);
    };
    // acc + element i as ONE fused multiply-add
    auto accumulate = [&](int i, VT acc) -> VT {
        if constexpr (SIGMA > 0)
            return fma_vt(mv[i], mx[i], acc);
        else
            return fma_vt(vt[i * OMEGA], gather(ct[i * OMEGA]), acc);
    };

    // Decode the descriptor and reduce the spill HERE, in the entry block, before any data-dependent
    // branch: values that are only consumed inside a branch get their loads sunk into it by the
    // compiler, which would serialise their round trip behind the tile's own.
    uint32_t flags = w0 << bit_all; // element i -> bit 31-i
    if (num_packet > 1)
        flags |= w1 >> (32 - bit_all);
    if constexpr (C16 || C31)
        flags = flags16;
    int y_off = (int)(w0 >> (32 - bit_y));
    const bool f0 = (flags >> 31) | (lane == 0);
    const bool present = f0 | ((flags & 0x7FFFFFFFu) != 0);
    if constexpr (C16 || C31) {
        // y_offset of the reference's descriptor (format_cuda.h:161-267) from the flags: segments that start in the lane,
        // exclusive wave prefix, minus one for lanes > 0 (as k_spmv_range does; equal to the stored field on every lane that
        // owns a flag)
        const int stop = __builtin_popcount(flags & 0x7FFFFFFFu);
        int segn = stop - (f0 ? 0 : 1) + (present ? 1 : 0);
        segn = segn > 0 ? segn : 0;
        const int incl = wave_scan_incl(segn);
        y_off = lane ? incl - segn - 1 : 0;
    }
    const unsigned long long pmask = __ballot(present);
    VT spill = 0;
    if constexpr (FUSED) {
        // lead_next is zero beyond the spill length, so a short spill needs only the first DPP steps
        const int L = ((mt.x >> 29) & 1u) ? (int)mt.z : 0;
        if (L > 0)
            spill = head_sum(lead_next, L);
    }
    const uint32_t rs_raw = __builtin_amdgcn_readfirstlane(tp0);
    const uint32_t row_stop = __builtin_amdgcn_readfirstlane(tp1) & ROW_MASK;

    if (rs_raw == row_stop) {
        // fast track: the whole tile lies inside one row (csr5_spmv_cuda.h:59-90)
        VT s = 0;
#pragma unroll
        for (int i = 0; i < sigma; i++)
            s = accumulate(i, s);
        s = wave_sum(s);
        if (lane == 0) {
            if constexpr (FUSED) // member of a multi-tile run: expected count lives at the run head
                carry_arrive(acc, cnt, calibrator, tile_ptr, (int)mt.y, meta[mt.y].x, t, false, s, y);
            else
                calibrator[t] = s;
        }
        return;
    }

    const bool empty_rows = (bool)(rs_raw >> 31);
    const int row_start = (int)(rs_raw & ROW_MASK);
    VT *y_local = y + row_start + 1;
    const int32_t *off_local = empty_rows ? offset + offset_ptr[t] : nullptr;

    constexpr bool LDSY = use_ldsy<VT, SIGMA, LDSY_REQ>();
    VT *seg = reinterpret_cast<VT *>(wave_lds);
    // store of the segment that owns slot `idx` of this tile's row range
    auto put = [&](int idx, VT v) {
        if constexpr (LDSY)
            seg[idx] = v;
        else
            y_local[empty_rows ? off_local[idx] : idx] = v;
    };
    bool direct = f0 && lane != 0;
    VT sum = product(0);
    VT first_sum = 0;
    int stored_hi = 0; // 1 + highest slot this lane has stored
#pragma unroll
    for (int i = 1; i < sigma; i++) {
        if ((flags >> (31 - i)) & 1u) {
            if (direct) {
                put(y_off, sum);
                stored_hi = y_off + 1;
            } else {
                first_sum = sum;
            }
            y_off += direct;
            direct = true;
            sum = 0;
        }
        sum = accumulate(i, sum);
    }
    if (!direct)
        first_sum = sum;

    // cross-lane step: every lane that owns a flag adds the leading partials of the lanes behind it,
    // up to and including the next lane that owns a flag:  S[l] = R[l+1],
    // R[j] = lead[j] + (present[j] ? 0 : R[j+1])  -- backward segmented scan, 6 shuffle steps.
    VT R = f0 ? (VT)0 : first_sum;
    // All on DPP / readlane (a ds_bpermute shuffle costs an LDS round trip per step): 4 in-row steps
    // (row_shl reads 0 across a 16-lane row edge), then the three row edges top-down: the lanes whose
    // run leaves row r add the finished R of the first lane of row r+1.  Steps no lane needs are skipped
    // by scalar tests on the flag-owner mask (a step of k lanes matters only if k absent lanes are adjacent).
    const unsigned long long z1 = ~pmask;
    if (z1) {
        const unsigned long long ahead = pmask >> lane;
        const int dist = ahead ? __builtin_ctzll(ahead) : OMEGA - 1 - lane;
        {
            const VT up = dpp_move<DPP_ROW_SHL1>(R);
            R += dist >= 1 ? up : (VT)0;
        }
        const unsigned long long z2 = z1 & (z1 >> 1);
        if (z2) {
            {
                const VT up = dpp_move<DPP_ROW_SHL2>(R);
                R += dist >= 2 ? up : (VT)0;
            }
            const unsigned long long z4 = z2 & (z2 >> 2);
            if (z4) {
                {
                    const VT up = dpp_move<DPP_ROW_SHL4>(R);
                    R += dist >= 4 ? up : (VT)0;
                }
                if (z4 & (z4 >> 4)) {
                    const VT up = dpp_move<DPP_ROW_SHL8>(R);
                    R += dist >= 8 ? up : (VT)0;
                }
            }
        }
        const int reach = lane + dist;
#pragma unroll
        for (int edge = 48; edge >= 16; edge -= 16) {
            if (!((pmask >> (edge - 1)) & 1ull)) { // lane edge-1 owns no flag: its run crosses the edge
                const VT carry_in = bcast_lane(R, edge);
                R += ((lane >> 4) == (edge >> 4) - 1 && reach >= edge) ? carry_in : (VT)0;
            }
        }
    }
    const VT S = lane_above(R); // lane 63 gets 0
    if (present)
        sum += S;

    const int last_present = 63 - __builtin_clzll(pmask);
    bool closing_to_protocol = false; // this lane's last segment goes to the arrival protocol
    if constexpr (FUSED) {
        const bool close_carry = (mt.x >> 30) & 1u;
        const bool close_local = (mt.x >> 29) & 1u;
        if (close_local && lane == last_present) // finish the closing row with its short spill
            sum += spill;
        closing_to_protocol = direct && close_carry && !close_local && lane == last_present;
    }
    if (direct && !closing_to_protocol) {
        put(y_off, sum);
        stored_hi = y_off + 1;
    }
    if constexpr (LDSY) {
        // flush: slots 0..nseg-1 are exactly the rows that start in this tile (minus a carried closing
        // row); slot indices grow with the lane, so the highest storing lane knows nseg
        const unsigned long long smask = __ballot(stored_hi != 0);
        if (smask) {
            const int nseg = __builtin_amdgcn_readlane(stored_hi, 63 - __builtin_clzll(smask));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // Two loops, not one with a select: the offset load would sit in the loop of every tile, and a store
            // whose address depends on a load makes the compiler drain the vector-memory counter -- i.e. the previous
            // iteration's STORE -- before it issues the next one (one store round trip per 64 segments).
            if (empty_rows) {
                for (int j = lane; j < nseg; j += OMEGA)
                    y_local[off_local[j]] = seg[j];
            }
            if (!empty_rows) {
                for (int j = lane; j < nseg; j += OMEGA)
                    y_local[j] = seg[j];
            }
        }
    }
    if constexpr (FUSED) {
        const bool lead_skip = (mt.x >> 28) & 1u;
        if (closing_to_protocol)
            carry_arrive(acc, cnt, calibrator, tile_ptr, t + 1, mt_next_x, t, true, sum, y);
        if (lane == 0 && !lead_skip) {
            const int slot = (int)mt.y;
            const uint32_t head_meta = slot == t ? mt.x : meta[slot].x;
            carry_arrive(acc, cnt, calibrator, tile_ptr, slot, head_meta, t, false, direct ? first_sum : sum, y);
        }
    } else {
        if (lane == 0)
            calibrator[t] = direct ? first_sum : sum;
    }
}

// One tile per wavefront, WAVES_PER_BLOCK tiles per workgroup; the CSR tail = extra workgroups of the same grid.
template <typename VT, int SIGMA, bool FUSED, bool XWIN, bool LDSY_REQ, bool NT = false, bool C16 = false, bool C31 = false>
__global__ void __launch_bounds__(BLOCK)
k_spmv(Geometry g, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
       const VT *__restrict__ val, const VT *__restrict__ x, const uint32_t *__restrict__ tile_ptr,
       const uint32_t *__restrict__ tile_desc, const int32_t *__restrict__ offset_ptr,
       const int32_t *__restrict__ offset, VT *__restrict__ calibrator, VT *__restrict__ y,
       int tile_blocks, int xcd_remap, VT *acc, uint32_t *cnt, const uint4 *__restrict__ meta,
       const uint32_t *__restrict__ hdr, const uint32_t *__restrict__ col16, const int32_t *__restrict__ base16,
       const uint32_t *__restrict__ col31)
{
    // Pull EVERY kernel argument into SGPRs with the first batch of scalar loads: an argument that is
    // first touched further down would otherwise cost its own kernarg round trip on the critical path.
    asm volatile("" ::"s"(row_ptr), "s"(col), "s"(val), "s"(x), "s"(tile_ptr), "s"(tile_desc),
                 "s"(offset_ptr), "s"(offset), "s"(calibrator), "s"(y), "s"(acc), "s"(cnt), "s"(meta), "s"(hdr),
                 "s"(g.nnz), "s"(g.p), "s"(g.m), "s"(g.sigma), "s"(g.tail_start), "s"(g.tile_elems),
                 "s"(g.bit_y), "s"(g.num_packet), "s"(tile_blocks), "s"(xcd_remap));
    // dynamic LDS: the tail's product buffer (T elements) or one x-window / y-segment region per wavefront
    extern __shared__ __attribute__((aligned(16))) char smem[];
    int blk = blockIdx.x;
    if (blk >= tile_blocks) {
        tail_rows<VT, SIGMA>(g, row_ptr, col, val, x, y, blk - tile_blocks, reinterpret_cast<VT *>(smem), [&](VT sum) {
            if constexpr (FUSED) {
                const uint4 mt = meta[g.p - 1];
                if (!((mt.x >> 28) & 1u)) // else tile p-2 already owns this row (short spill)
                    carry_arrive(acc, cnt, calibrator, tile_ptr, (int)mt.y, meta[mt.y].x, g.p - 1, false, sum, y);
            } else {
                calibrator[g.p - 1] = sum;
            }
        });
        return;
    }
    if (xcd_remap) {
        // workgroup b runs on XCD b % 8 (observed dispatch order; used for L2 locality only):
        // give every XCD one contiguous range of tiles instead of every 8th workgroup.
        const int q = tile_blocks / NUM_XCD, rem = tile_blocks % NUM_XCD;
        const int xcd = blk % NUM_XCD;
        blk = xcd * q + (xcd < rem ? xcd : rem) + blk / NUM_XCD;
    }
    const int lane = threadIdx.x & (OMEGA - 1);
    const int t = __builtin_amdgcn_readfirstlane(blk * WAVES_PER_BLOCK + (int)(threadIdx.x >> 6));
    if (t >= g.p - 1)
        return;
    TILE_STAMP(0, false);
    tile_body<VT, SIGMA, FUSED, XWIN, LDSY_REQ, NT, C16, C31>(
        g, t, lane, col, val, x, tile_ptr, tile_desc, offset_ptr, offset, calibrator, y, acc, cnt, meta, hdr,
        smem + (threadIdx.x >> 6) * wave_lds_bytes<VT, SIGMA, XWIN, LDSY_REQ>(), col16, base16, col31);
    TILE_STAMP(2, false);
}

// ---- carry resolution by a second launch ---------------------------------------------------------
// One thread per run head (carry_meta[t].y == t).
//   LONG_ONLY = false (two-pass mode): every run.  The first carry of a row that begins exactly on a tile
//     boundary stores, otherwise the closing partial that k_spmv stored into y[r] comes first (CSR5_avx2
//     csr5_spmv_avx2.h:42-49,284-291 semantics).
//   LONG_ONLY = true (fused mode, launched only when the matrix has such rows): only the runs longer than
//     RUN_SERIAL_MAX tiles, whose parties parked their partials (closing partial in acc[head]).
// Summation order = sum_run(), identical in both modes and to the fused kernel's in-launch finisher; long
// runs are summed by the whole wavefront.
template <typename VT, bool LONG_ONLY>
__global__ void __launch_bounds__(BLOCK)
k_calibrate(Geometry g, const uint32_t *__restrict__ tile_ptr, const uint4 *__restrict__ meta,
            const VT *__restrict__ calibrator, const VT *__restrict__ acc, VT *__restrict__ y)
{
    const int lane = threadIdx.x & (OMEGA - 1);
    const int t = blockIdx.x * BLOCK + threadIdx.x;
    bool head = false;
    int len = 0;
    bool has_first = false;
    int r = 0;
    // deferred carries make this launch part of every SpMV of some matrices: the words a run of <= 2 parked partials needs
    // (nearly all of them: a row cut once or twice) are requested WITH the meta word instead of after it -- one round trip less
    VT spec_first = 0, spec0 = 0, spec1 = 0;
    if (LONG_ONLY && t < g.p) {
        spec_first = acc[t];
        spec0 = calibrator[t];
        spec1 = calibrator[t + 1 < g.p ? t + 1 : t];
    }
    if (t < g.p) {
        const uint4 mt = meta[t];
        head = (int)mt.y == t;
        r = (int)(tile_ptr[t] & ROW_MASK);
        if ((mt.x >> 28) & 1u) { // short-spill head (fused-mode classification): one carry onto y[r]
            len = 1;
            has_first = true;
        } else {
            has_first = (mt.x >> 27) & 1u;
            len = (int)(mt.x & 0x00FFFFFFu) - (has_first ? 1 : 0);
        }
        head = head && r < g.m && len > 0;
        if (LONG_ONLY)
            head = head && ((mt.x >> 26) & 1u);
    }
    // (LONG_ONLY also serves the runs whose parties were told to park although they are short: deferred carries,
    //  CSR5HIP_OPT_DEFER_CARRIES -- the closing partial waits in acc[head], as for a long run)
    if (LONG_ONLY && head && len <= 2) {
        VT total = has_first ? spec_first + spec0 : spec0; // (sum_run's association)
        if (len == 2)
            total += spec1;
        y[r] = total;
    } else if (head && len <= RUN_SERIAL_MAX)
        y[r] = sum_run<VT, false>(calibrator, t, len, has_first, has_first ? (LONG_ONLY ? acc[t] : y[r]) : (VT)0, lane, lane);
    unsigned long long todo = __ballot(head && len > RUN_SERIAL_MAX);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int slot = __shfl(t, leader, OMEGA);
        const int ln = __shfl(len, leader, OMEGA);
        const bool hf = __shfl((int)has_first, leader, OMEGA);
        const int row = __shfl(r, leader, OMEGA);
        VT first = 0;
        if (hf && lane == leader)
            first = LONG_ONLY ? acc[slot] : y[row];
        const VT total = sum_run<VT, false>(calibrator, slot, ln, hf, first, lane, leader);
        if (lane == leader)
            y[row] = total;
    }
}

// ---- dispatch ------------------------------------------------------------------------------------
template <typename VT, int SIGMA, bool FUSED, bool XWIN, bool LDSY_REQ, bool NT = false, bool C16 = false, bool C31 = false>
static hipError_t launch_one(const Geometry &g, const DeviceArrays &d, const void *x, void *y,
                             const SpmvOptions &opt, hipStream_t s)
{
    const int tile_blocks = g.p > 1 ? (g.p - 1 + WAVES_PER_BLOCK - 1) / WAVES_PER_BLOCK : 0;
    const int tail_rows_n = g.m - g.tail_start;
    const int tail_blocks = tail_rows_n > 0 ? (tail_rows_n + BLOCK - 1) / BLOCK : 0;
    if (tile_blocks + tail_blocks == 0)
        return hipSuccess;
    size_t lds = (size_t)g.tile_elems * sizeof(VT); // tail product buffer (tail workgroups)
    if (lds < (size_t)WAVES_PER_BLOCK * wave_lds_bytes<VT, SIGMA, XWIN, LDSY_REQ>())
        lds = (size_t)WAVES_PER_BLOCK * wave_lds_bytes<VT, SIGMA, XWIN, LDSY_REQ>();
    hipLaunchKernelGGL((k_spmv<VT, SIGMA, FUSED, XWIN, LDSY_REQ, NT, C16, C31>), dim3(tile_blocks + tail_blocks), dim3(BLOCK), lds, s,
                       g, d.row_ptr, d.col, (const VT *)d.val, (const VT *)x, d.tile_ptr,
                       d.tile_desc, d.offset_ptr, d.offset, (VT *)d.calibrator, (VT *)y, tile_blocks,
                       opt.xcd_remap, (VT *)d.carry_acc, d.carry_cnt,
                       reinterpret_cast<const uint4 *>(d.carry_meta), d.tile_hdr, d.col16, d.base16, d.col31);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || (FUSED && !opt.long_runs))
        return e;
    // second launch: two-pass mode always; fused mode only when some row spans > RUN_SERIAL_MAX tiles
    hipLaunchKernelGGL((k_calibrate<VT, FUSED>), dim3((g.p + BLOCK - 1) / BLOCK), dim3(BLOCK), 0, s, g,
                       d.tile_ptr, reinterpret_cast<const uint4 *>(d.carry_meta), (const VT *)d.calibrator,
                       (const VT *)d.carry_acc, (VT *)y);
    return hipGetLastError();
}

template <typename VT, bool FUSED>
static hipError_t launch_sigma(const Geometry &g, const DeviceArrays &d, const void *x, void *y,
                               const SpmvOptions &opt, hipStream_t s)
{
    switch (g.sigma) {
#define CSR5_CASE(S)                                                                               \
    case S:                                                                                        \
        if constexpr (FUSED) {                                                                     \
            if constexpr (col16_sigma(S)) {                                                        \
                if (opt.x_window && opt.col16 && d.col16)                                          \
                    return opt.lds_y ? launch_one<VT, S, FUSED, true, true, false, true>(g, d, x, y, opt, s)  \
                                     : launch_one<VT, S, FUSED, true, false, false, true>(g, d, x, y, opt, s); \
            }                                                                                      \
            if (opt.x_window)                                                                      \
                return opt.lds_y ? launch_one<VT, S, FUSED, true, true>(g, d, x, y, opt, s)        \
                                 : launch_one<VT, S, FUSED, true, false>(g, d, x, y, opt, s);      \
            if constexpr (col31_sigma(S)) {                                                        \
                if (opt.col31 && d.col31) {                                                        \
                    if (opt.stream_nt)                                                             \
                        return opt.lds_y ? launch_one<VT, S, FUSED, false, true, true, false, true>(g, d, x, y, opt, s)   \
                                         : launch_one<VT, S, FUSED, false, false, true, false, true>(g, d, x, y, opt, s); \
                    return opt.lds_y ? launch_one<VT, S, FUSED, false, true, false, false, true>(g, d, x, y, opt, s)      \
                                     : launch_one<VT, S, FUSED, false, false, false, false, true>(g, d, x, y, opt, s);    \
                }                                                                                  \
            }                                                                                      \
            if (opt.stream_nt)                                                                     \
                return opt.lds_y ? launch_one<VT, S, FUSED, false, true, true>(g, d, x, y, opt, s) \
                                 : launch_one<VT, S, FUSED, false, false, true>(g, d, x, y, opt, s); \
        }                                                                                          \
        return opt.lds_y ? launch_one<VT, S, FUSED, false, true>(g, d, x, y, opt, s)               \
                         : launch_one<VT, S, FUSED, false, false>(g, d, x, y, opt, s);
#if defined(CSR5_FEW_SIGMAS) // experiment builds: few instantiations only
        CSR5_CASE(4) CSR5_CASE(5) CSR5_CASE(6) CSR5_CASE(8) CSR5_CASE(12) CSR5_CASE(16) CSR5_CASE(20) CSR5_CASE(24) CSR5_CASE(32)
#else
        CSR5_CASE(4) CSR5_CASE(5) CSR5_CASE(6) CSR5_CASE(7) CSR5_CASE(8) CSR5_CASE(9) CSR5_CASE(10)
        CSR5_CASE(11) CSR5_CASE(12) CSR5_CASE(13) CSR5_CASE(14) CSR5_CASE(15) CSR5_CASE(16)
        CSR5_CASE(17) CSR5_CASE(18) CSR5_CASE(19) CSR5_CASE(20) CSR5_CASE(21) CSR5_CASE(22)
        CSR5_CASE(23) CSR5_CASE(24) CSR5_CASE(25) CSR5_CASE(26) CSR5_CASE(27) CSR5_CASE(28)
        CSR5_CASE(29) CSR5_CASE(30) CSR5_CASE(31) CSR5_CASE(32)
#endif
#undef CSR5_CASE
    default: return launch_one<VT, 0, FUSED, false, false>(g, d, x, y, opt, s);
    }
}

// The product build compiles this file twice (-DCSR5_SPMV_ONLY_F64 / -DCSR5_SPMV_ONLY_F32: the two halves of the
// ~460 kernel instantiations build in parallel); experiment builds compile it once with neither macro.
#if !defined(CSR5_SPMV_ONLY_F32)
hipError_t launch_spmv_f64(const Geometry &g, const DeviceArrays &d, const void *x, void *y,
                           const SpmvOptions &opt, hipStream_t s)
{
    return opt.mode == 1 ? launch_sigma<double, true>(g, d, x, y, opt, s)
                         : launch_sigma<double, false>(g, d, x, y, opt, s);
}
#endif
#if !defined(CSR5_SPMV_ONLY_F64)
hipError_t launch_spmv_f32(const Geometry &g, const DeviceArrays &d, const void *x, void *y,
                           const SpmvOptions &opt, hipStream_t s)
{
    return opt.mode == 1 ? launch_sigma<float, true>(g, d, x, y, opt, s)
                         : launch_sigma<float, false>(g, d, x, y, opt, s);
}
#endif

#if !defined(CSR5_SPMV_ONLY_F32)
hipError_t launch_spmv_f32(const Geometry &g, const DeviceArrays &d, const void *x, void *y,
                           const SpmvOptions &opt, hipStream_t s);

hipError_t launch_spmv(const Geometry &g, const DeviceArrays &d, int value_type, const void *x,
                       void *y, const SpmvOptions &opt, hipStream_t s)
{
    if (g.p <= 0)
        return hipSuccess;
    if (opt.hot) // column words are hot-encoded: only the persistent range kernel understands them (csr5_hot.hip)
        return opt.mode == 1 ? launch_spmv_hot(g, d, value_type, x, y, opt, s) : hipErrorInvalidValue;
    return value_type == CSR5HIP_F64 ? launch_spmv_f64(g, d, x, y, opt, s) : launch_spmv_f32(g, d, x, y, opt, s);
}
#endif

} // namespace csr5

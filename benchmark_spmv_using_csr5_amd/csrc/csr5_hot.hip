// csr5_hot.hip -- the persistent kernel of the column-slab child with an LDS hot table (csr5_slab.hip), rounds 3-4.
//
// What it computes is the reference's tile kernel (CSR5_cuda/detail/cuda/csr5_spmv_cuda.h:59-311: fast / normal track,
// lane-local flag walk, cross-lane segmented sum) on the child's CSR5 arrays; how the work is laid out is ours:
//   * one workgroup of HOT_WAVES wavefronts per CU stays resident; XCD x walks its slabs in order, refilling the LDS table
//     with the slab's hot x entries (hot gathers = ds_read_b64, cold ones = range-checked buffer loads from the permuted
//     copy of x);
//   * every WAVEFRONT owns ONE CONTIGUOUS RANGE of the slab's tiles (HOT_RANGES_PER_SLAB = 256 ranges per slab).  The row that is open at a
//     tile boundary therefore meets its continuation in the registers of the same wavefront: a tile needs no header,
//     no re-read of its successor's first elements, no carry slot and no atomic -- the reference's calibrate pass
//     (csr5_spmv_cuda.h:313-382) shrinks to ONE leading partial per range (`lead`), added by k_range_finish;
//   * a tile's finished partial sums are compacted in LDS and leave as one contiguous run P[row_start ...]: the
//     stores of a wavefront are sequential over its whole range.
// Round 2's form (tiles dealt round robin, per-tile 32-byte header, short-spill re-reads, arrival protocol) is gone.
#include "csr5_internal.h"
#include "csr5_slabmap.h"
#include "csr5_wave.h"

#include <type_traits>

namespace csr5 {

constexpr int HOT_BLOCK = HOT_WAVES * OMEGA;

// ---- which tiles of a slab does wavefront range rho walk? ------------------------------------------------------------------
// The slab's n tiles are dealt evenly to the workgroups (HOT_WAVES ranges each); inside a workgroup the wavefronts do NOT get
// equal shares.  With two wavefronts per SIMD the one that was launched first wins every issue conflict: measured with
// wall-clock stamps per range (scripts/experiments/round5/range_stamps.py, R-MAT 24) wavefronts 0-3 of a workgroup finish
// equal ranges 5.5 % below the mean and wavefronts 4-7 5.5 % above it, in every workgroup and slab (with a skew of 80 per mille: 0.98 ... 1.016) -- and the workgroup waits
// for its slowest wavefront at every slab boundary (the table refill).  The first half of a workgroup's wavefronts therefore
// takes HOT_SKEW_PERMIL per mille more tiles than the mean, the second half as much less.
#ifndef CSR5_HOT_SKEW_PERMIL
#define CSR5_HOT_SKEW_PERMIL 85
#endif
constexpr int HOT_SKEW_PERMIL = HOT_WAVES == 8 ? CSR5_HOT_SKEW_PERMIL : 0;
__host__ __device__ __forceinline__ int hot_range_begin(int n, int rho) // first tile (relative to the slab) of range rho; rho == ranges: n
{
    constexpr int nwg = HOT_RANGES_PER_SLAB / HOT_WAVES;
    const int wg = rho / HOT_WAVES, j = rho % HOT_WAVES;
    if (wg >= nwg)
        return n;
    const int q = n / nwg, rem = n % nwg;
    const int wb = wg * q + (wg < rem ? wg : rem), sz = q + (wg < rem ? 1 : 0);
    // cumulative share of wavefronts 0 .. j-1 in 1/(1000 HOT_WAVES): 1000 + skew each in the first half, 1000 - skew in the second
    constexpr int half = HOT_WAVES / 2;
    const int cum = j <= half ? j * (1000 + HOT_SKEW_PERMIL) : half * (1000 + HOT_SKEW_PERMIL) + (j - half) * (1000 - HOT_SKEW_PERMIL);
    return wb + (int)((long long)sz * cum / (1000 * HOT_WAVES));
}

// The child's column words come as 3-byte codes (k_hot_encode ENC_PACK) in CSR order -- lane l's sigma codes are
// consecutive: 2 sigma bytes of col_lo, sigma bytes of col_hi (whole dwords: the child's sigma is a multiple of four) --
// and are decoded into c[] when they have arrived.
// ST = the type the values are STORED in (VT, or float for an fp64 matrix whose values are all exactly representable in fp32:
// CSR5HIP_OPT_NARROW_VALUES -- half the value stream, the same products bit for bit)
template <typename ST, int SIGMA>
struct TileRegs {
    static_assert(SIGMA % 4 == 0, "a lane's column codes are whole dwords");
    int32_t c[SIGMA];
    uint32_t plo[SIGMA / 2], phi[SIGMA / 4];
    ST v[SIGMA];
    uint32_t flags; // bit 31 - i = element i starts a row (bit 22 of its column code; set by decode)
    uint32_t tp0, tp1;
};

typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
// N consecutive dwords (p 4 N-byte aligned ... at least as far as the widest piece used) with the widest loads
template <int N, bool NT>
__device__ __forceinline__ void load_dwords(uint32_t *dst, const uint32_t *p)
{
    if constexpr (N >= 4) {
        const u32x4 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x4 *>(p)) : *reinterpret_cast<const u32x4 *>(p);
        dst[0] = v.x, dst[1] = v.y, dst[2] = v.z, dst[3] = v.w;
        load_dwords<N - 4, NT>(dst + 4, p + 4);
    } else if constexpr (N >= 2) {
        const u32x2 v = NT ? __builtin_nontemporal_load(reinterpret_cast<const u32x2 *>(p)) : *reinterpret_cast<const u32x2 *>(p);
        dst[0] = v.x, dst[1] = v.y;
        load_dwords<N - 2, NT>(dst + 2, p + 2);
    } else if constexpr (N == 1) {
        dst[0] = NT ? __builtin_nontemporal_load(p) : *p;
    }
}

// every load of tile t: column codes first (the gathers wait for them only; they carry the tile's bit flags too: the
// descriptor array of the reference format is not read), then the tile_ptr pair (scalar) and the values
// PER = values per 16-byte piece of the STORED type: the child's values lie in lane-major 16-byte pieces (k_transpose_values;
// the fp32 copy of CSR5HIP_OPT_NARROW_VALUES in pieces of four: k_narrow)
template <typename VT, int SIGMA, bool NT, int PER>
__device__ __forceinline__ void range_load(TileRegs<VT, SIGMA> &r, const uint16_t *__restrict__ col_lo,
                                           const uint8_t *__restrict__ col_hi, const VT *__restrict__ val,
                                           const uint32_t *__restrict__ tile_ptr, int t, int lane)
{
    constexpr int T = OMEGA * SIGMA;
    typedef VT piece_t __attribute__((ext_vector_type(PER)));
    const piece_t *vp = reinterpret_cast<const piece_t *>(val + (size_t)t * T) + lane;
    const size_t first = (size_t)t * T + (size_t)lane * SIGMA;
    load_dwords<SIGMA / 2, NT>(r.plo, reinterpret_cast<const uint32_t *>(col_lo + first));
    load_dwords<SIGMA / 4, NT>(r.phi, reinterpret_cast<const uint32_t *>(col_hi + first));
    {
        // the tile_ptr pair is wave-uniform: through the scalar cache (constant address space -> s_load_dwordx2, requested
        // one tile ahead like the streams), which keeps it off the vector memory path -- the unit this kernel saturates
        const auto *tpc = (const __attribute__((address_space(4))) uint32_t *)(uintptr_t)tile_ptr;
        r.tp0 = tpc[t];
        r.tp1 = tpc[t + 1];
    }
#pragma unroll
    for (int q = 0; q < SIGMA / PER; q++) {
        const piece_t w = NT ? __builtin_nontemporal_load(vp + q * OMEGA) : vp[q * OMEGA];
#pragma unroll
        for (int e = 0; e < PER; e++)
            r.v[q * PER + e] = w[e];
    }
}

// State of the row that is open at the current tile boundary (wave-uniform).
template <typename VT>
struct OpenRow {
    int row;      // child row (segment) of the open partial
    VT val;       // its partial sum so far (same value in every lane)
    bool is_lead; // the row was already open when the range began: the partial goes to lead[range], not to P
};

#if defined(CSR5_RANGE_STAMPS) // experiment builds only: wall-clock stamps (100 MHz) of every wavefront range: start, end
__device__ unsigned long long g_range_stamps[2 * 64 * HOT_RANGES_PER_SLAB];
extern "C" int csr5hip_debug_range_stamps(unsigned long long *dst, int count)
{
    return (int)hipMemcpyFromSymbol(dst, HIP_SYMBOL(g_range_stamps), (size_t)count * sizeof(unsigned long long));
}
#endif

template <typename VT, int SIGMA, bool NT, int DEPTH, typename ST = VT>
__global__ void __launch_bounds__(HOT_BLOCK)
k_spmv_range(Geometry g, const ST *__restrict__ val, const uint32_t *__restrict__ tile_ptr, VT *__restrict__ P,
             VT *__restrict__ lead, HotParams hp)
{
    static_assert(num_packet_of(SIGMA) == 1, "the bit flags of a hot child's tile fit one word per lane");
    using word_t = typename std::conditional<sizeof(VT) == 8, unsigned long long, unsigned>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // typed LDS pointer: keeps the table reads on ds_read (a generic pointer would merge the hot/cold select into one
    // flat_load)
    auto *hot = (__attribute__((address_space(3))) VT *)(smem);
    const int xcd = blockIdx.x % NUM_XCD, wg = blockIdx.x / NUM_XCD, nwg = gridDim.x / NUM_XCD;
    const int lane = threadIdx.x & (OMEGA - 1), wave = threadIdx.x >> 6;
    VT *seg = reinterpret_cast<VT *>(smem + (size_t)hp.capacity * sizeof(VT) + (size_t)wave * HOT_WAVE_LDS);
    // Cold gathers come from the cold region of the permuted copy of x (hp.xp: [slabs][capacity] table images, then every
    // slab's cold columns in descending order of use -- k_x_permute), so that the part of x a slab gathers from is dense and
    // its popular prefix stays in the XCD's L2.
    const VT *xcold = static_cast<const VT *>(hp.xp) + (size_t)hp.slabs * hp.capacity;
    const auto xbuf = __builtin_amdgcn_make_buffer_rsrc(const_cast<VT *>(xcold), (short)0, hp.cold_total * (int)sizeof(VT), 0x00020000);

    for (int r = 0; r < hp.rounds; r++) {
        const int k = hp.tile0[hp.slabs + 1 + xcd * hp.rounds + r];
        const int nhot = hp.count[k];
        __syncthreads(); // every wavefront is done with the previous slab's table
        {
            // the slab's table image is one contiguous run of the permuted copy: a coalesced copy
            const VT *img = static_cast<const VT *>(hp.xp) + (size_t)k * hp.capacity;
            for (int j0 = 0; j0 < nhot; j0 += HOT_BLOCK * 8) {
                VT xw[8];
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int j = j0 + q * HOT_BLOCK + (int)threadIdx.x;
                    xw[q] = img[j < nhot ? j : 0];
                }
#pragma unroll
                for (int q = 0; q < 8; q++) {
                    const int j = j0 + q * HOT_BLOCK + (int)threadIdx.x;
                    if (j < nhot)
                        hot[j] = j ? xw[q] : (VT)0; // slot 0 = +0.0: what the cold lanes read
                }
            }
        }
        __syncthreads();

        // this wavefront's contiguous range of the slab's tiles
        const int t0 = hp.tile0[k], n = hp.tile0[k + 1] - t0;
        const int nr = nwg * (HOT_BLOCK / OMEGA), rho = wg * (HOT_BLOCK / OMEGA) + wave;
        const int tb = __builtin_amdgcn_readfirstlane(t0 + hot_range_begin(n, rho));
        const int te = __builtin_amdgcn_readfirstlane(t0 + hot_range_begin(n, rho + 1));
        VT *const my_lead = lead + (size_t)k * nr + rho;
#if defined(CSR5_RANGE_STAMPS)
        if (lane == 0)
            g_range_stamps[2 * (k * nr + rho)] = wall_clock64();
#endif
        if (tb >= te) {
            if (lane == 0)
                *my_lead = 0;
            continue;
        }

        OpenRow<VT> open{-1, (VT)0, true};
        // the partial of the open row is complete (its row ended): to P, or -- the row was open when the range began --
        // to this range's lead word
        auto emit_open = [&]() {
            if (lane == 0)
                *(open.is_lead ? my_lead : P + open.row) = open.val;
        };
        // One x gather per element, branch-free: cold lanes (plain column word) read x through a raw buffer load; hot
        // lanes (bit 31 set) carry the byte offset 0xFFFFFFFF there, which the buffer's range check turns into "return
        // 0, touch no memory".  Every lane then reads the LDS table (cold lanes slot 0 = +0.0) and a bitwise OR merges
        // the two words exactly.
        auto cold_word = [&](int32_t cw) -> word_t {
            const unsigned off = cw < 0 ? 0xFFFFFFFFu : (unsigned)cw * (unsigned)sizeof(VT);
            if constexpr (sizeof(VT) == 8)
                return __builtin_bit_cast(word_t, __builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 0));
            else
                return __builtin_bit_cast(word_t, __builtin_amdgcn_raw_buffer_load_b32(xbuf, off, 0, 0));
        };
        auto table_word = [&](int32_t cw) -> word_t {
            return __builtin_bit_cast(word_t, hot[cw < 0 ? (unsigned)cw & 0x7FFFFFFFu : 0u]);
        };

        // the 3-byte codes of tile t -> gather words (bit 31 | slot, or the index into the cold region of the
        // permuted copy: start of the element's slab + its column's rank there).  The element's slab is this slab, except
        // behind the slab's end inside its last tile (the elements there belong to the following slab(s)).  A quarter fewer
        // column bytes and two wide loads instead of eight; the decode is a shift, an or and an add per element.
        const long long own_end = (long long)hp.slab_off[k + 1];
        const int32_t own_base = hp.cold_base[k];
        auto decode = [&](TileRegs<ST, SIGMA> &tr, int t) {
            {
                constexpr int T = OMEGA * SIGMA;
                const long long first = (long long)t * T;
                uint32_t code[SIGMA];
                uint32_t fl = 0;
#pragma unroll
                for (int i = 0; i < SIGMA; i++) {
                    code[i] = ((tr.plo[i / 2] >> (16 * (i & 1))) & 0xFFFFu) | (((tr.phi[i / 4] >> (8 * (i & 3))) & 0xFFu) << 16);
                    fl |= ((code[i] >> 22) & 1u) << (31 - i); // the element starts a row (the reference's bit flag)
                    code[i] &= 0xBFFFFFu;
                }
                tr.flags = fl;
                if (first + T <= own_end) { // (wave-uniform) straight-line: this code sits in front of the tile's gathers
#pragma unroll
                    for (int i = 0; i < SIGMA; i++)
                        tr.c[i] = (code[i] & 0x800000u) ? (int32_t)(0x80000000u | (code[i] & 0x3FFFFFu))
                                                        : (int32_t)code[i] + own_base;
                } else { // the slab ends inside this tile (one tile per slab)
#pragma unroll
                    for (int i = 0; i < SIGMA; i++) {
                        const long long pos = first + (long long)lane * SIGMA + i;
                        int32_t base = own_base;
                        for (int j = k + 1; j < hp.slabs; j++)
                            base = pos >= (long long)hp.slab_off[j] ? hp.cold_base[j] : base;
                        tr.c[i] = (code[i] & 0x800000u) ? (int32_t)(0x80000000u | (code[i] & 0x3FFFFFu)) : (int32_t)code[i] + base;
                    }
                }
            }
        };

        // ---- one tile whose loads (streams in `tr`, cold gathers in `xg`) are in flight or done -------------------------
        auto compute = [&](const TileRegs<ST, SIGMA> &tr, const word_t (&xg)[SIGMA]) {
            VT mx[SIGMA];
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                mx[i] = __builtin_bit_cast(VT, (word_t)(xg[i] | table_word(tr.c[i])));
            const uint32_t tp0 = __builtin_amdgcn_readfirstlane(tr.tp0), tp1 = __builtin_amdgcn_readfirstlane(tr.tp1);
            const int rs = (int)(tp0 & ROW_MASK);
            const uint32_t flags = tr.flags; // element i -> bit 31-i
            const bool f0 = (flags >> 31) | (lane == 0);
            const bool present = f0 | ((flags & 0x7FFFFFFFu) != 0);
            // y_offset of the reference's descriptor (format_cuda.h:161-267, our k_tile_desc), recomputed from the flags instead
            // of loaded: segments that start in the lane, exclusive wave prefix, minus one for lanes > 0
            int y_off;
            {
                const int stop = __builtin_popcount(flags & 0x7FFFFFFFu);
                int segn = stop - (f0 ? 0 : 1) + (present ? 1 : 0);
                segn = segn > 0 ? segn : 0;
                const int incl = wave_scan_incl(segn); // DPP, no LDS crossbar trips
                y_off = lane ? incl - segn - 1 : 0;
            }
            if (open.row < 0)
                open.row = rs; // first tile of the range
            if (rs != open.row) {
                // the tile begins with a new row: the open one ended exactly on the boundary
                emit_open();
                open.row = rs;
                open.val = 0;
                open.is_lead = false;
            }
            if (tp0 == tp1) {
                // fast track: the whole tile lies inside the open row (csr5_spmv_cuda.h:59-90)
                VT s = 0;
#pragma unroll
                for (int i = 0; i < SIGMA; i++)
                    s = fma_vt((VT)tr.v[i], mx[i], s);
                open.val += wave_sum(s);
                return;
            }
            const unsigned long long pmask = __ballot(present);
            bool direct = f0 && lane != 0;
            VT sum = (VT)tr.v[0] * mx[0];
            VT first_sum = 0;
#pragma unroll
            for (int i = 1; i < SIGMA; i++) {
                if ((flags >> (31 - i)) & 1u) {
                    if (direct)
                        seg[y_off] = sum;
                    else
                        first_sum = sum;
                    y_off += direct;
                    direct = true;
                    sum = 0;
                }
                sum = fma_vt((VT)tr.v[i], mx[i], sum);
            }
            if (!direct)
                first_sum = sum;
            // cross-lane step: backward segmented scan R[j] = lead[j] + (present[j] ? 0 : R[j+1]) on DPP row shifts and
            // v_readlane row carries; steps no lane needs are skipped by scalar tests on the flag-owner mask
            VT R = f0 ? (VT)0 : first_sum;
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
            // leading run of the tile (elements before the first row start at position >= 1): continues the open row
            const VT leading = bcast_lane(direct ? first_sum : sum, 0);
            const unsigned long long dmask = __ballot(direct);
            if (!dmask) {
                open.val += leading; // no row starts inside the tile
                return;
            }
            // Rows rs .. rs + nslot: row rs (the open row) is complete now, slots 0..nslot-2 are rows that start and end
            // inside the tile, slot nslot-1 -- the last segment of the highest flag-owning lane -- stays open.
            const int last = 63 - __builtin_clzll(pmask);
            const int nslot = __builtin_amdgcn_readlane(y_off, last) + 1;
            const VT closing = bcast_lane(sum, last);
            if (direct && lane != last)
                seg[y_off] = sum;
            const VT done = open.val + leading;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            VT *const out = P + rs;
            // one contiguous run P[rs .. rs + nslot): element 0 = the finished open row (or this range's lead word)
            for (int j = lane; j < nslot; j += OMEGA) {
                const VT vj = j == 0 ? done : seg[j > 0 ? j - 1 : 0];
                if (j > 0 || !open.is_lead)
                    out[j] = vj;
            }
            if (open.is_lead && lane == 0)
                *my_lead = done;
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // the next tile's LDS writes stay behind these reads
            __builtin_amdgcn_wave_barrier();
            open.row = rs + nslot;
            open.val = closing;
            open.is_lead = false;
        };

        auto load = [&](TileRegs<ST, SIGMA> &tr, int t) {
            range_load<ST, SIGMA, NT, 16 / (int)sizeof(ST)>(tr, hp.col_lo, hp.col_hi, val, tile_ptr, t, lane);
        };
        if constexpr (DEPTH == 1) {
            for (int t = tb; t < te; t++) {
                TileRegs<ST, SIGMA> a;
                word_t xa[SIGMA];
                load(a, t);
                __builtin_amdgcn_sched_barrier(0);
                decode(a, t);
#pragma unroll
                for (int i = 0; i < SIGMA; i++)
                    xa[i] = cold_word(a.c[i]);
                __builtin_amdgcn_sched_barrier(0);
                compute(a, xa);
            }
        } else {
            // DEPTH 2: the next tile's streams go out right in front of this tile's gathers (their long HBM round trip
            // starts first; same-call A/B: 1 % faster than behind them) and are in flight while it computes.  (Streams
            // TWO tiles ahead -- three register sets, 132 VGPRs -- measured equal at 8 wavefronts per CU and 3 % slower at
            // 16: profiles/r04_probes.txt.)  Tiles are taken in pairs (two register sets, no copies) and an odd last
            // tile is peeled: a `break` in the middle of the pair loop leaves the compiler a path on which the second set's loads are
            // still pending at the loop head, and it then drains the whole queue (s_waitcnt vmcnt(0)) in front of
            // every pair's gathers.
            TileRegs<ST, SIGMA> a, b;
            word_t xa[SIGMA];
            load(a, tb);
            int t = tb;
            for (; t + 1 < te; t += 2) {
                __builtin_amdgcn_sched_barrier(0);
                load(b, t + 1);
                __builtin_amdgcn_sched_barrier(0);
                decode(a, t);
#pragma unroll
                for (int i = 0; i < SIGMA; i++)
                    xa[i] = cold_word(a.c[i]);
                __builtin_amdgcn_sched_barrier(0);
                compute(a, xa);
                __builtin_amdgcn_sched_barrier(0);
                load(a, t + 2 < te ? t + 2 : t + 1);
                __builtin_amdgcn_sched_barrier(0);
                decode(b, t + 1);
#pragma unroll
                for (int i = 0; i < SIGMA; i++)
                    xa[i] = cold_word(b.c[i]);
                __builtin_amdgcn_sched_barrier(0);
                compute(b, xa);
            }
            if (t < te) {
                __builtin_amdgcn_sched_barrier(0);
                decode(a, t);
#pragma unroll
                for (int i = 0; i < SIGMA; i++)
                    xa[i] = cold_word(a.c[i]);
                __builtin_amdgcn_sched_barrier(0);
                compute(a, xa);
            }
        }
        emit_open(); // the last open row of the range: a later range may continue it (its lead is added by k_range_finish)
#if defined(CSR5_RANGE_STAMPS)
        if (lane == 0)
            g_range_stamps[2 * (k * nr + rho) + 1] = wall_clock64();
#endif
    }
}

// ---- after the persistent kernel: the CSR tail tile and the range seams -------------------------------------------------
// One thread per range R (and one for the CSR tail, "range" S * ranges_per_slab), 256 per workgroup.
// Tail (csr5_spmv_cuda.h:384-419): rows tail_start .. m-1, at most T <= 1024 non-zeros and (no empty rows in a slab child)
//   at most as many rows.  Every workgroup multiplies the tail's elements into LDS (one round trip, a few hundred
//   elements) so that it knows the partial of the first tail row -- the "lead" of the tail -- without waiting for another
//   workgroup; workgroup 0 also stores the other tail rows.
// Seams: for every row that is open at the start of a range (or of the tail), in range order:
//   P[row] = (the partial stored by the range in which the row begins, unless it begins exactly with this range)
//          + the leads of all consecutive ranges that start inside the row -- the reference's calibrate pass
//   (csr5_spmv_cuda.h:313-382) with a fixed association, so results are bit-reproducible.  Which row a range starts in is
//   known at conversion (k_range_heads: `head`, one word per range, non-decreasing), so the thread of the FIRST range of a
//   row finds the end of the row's run with one look at its neighbour (the usual case: the run is this range alone) or a
//   bisection of `head`, and adds the run's leads with independent loads; a run longer than RUN_WAVE ranges (a dense
//   row covers all 256 ranges of every slab) is summed by the whole wavefront.
constexpr uint32_t RANGE_EXACT = 0x80000000u; // head bit: the range's first row BEGINS with the range's first element
constexpr uint32_t RANGE_NONE = 0x7FFFFFFFu;  // head row of the (empty) ranges behind the last tile
constexpr int RUN_WAVE = 64;

struct RangeHead {
    int row;            // first row of the range, -1 = the range holds no tile
    long long first;    // index of its first element
};
__device__ __forceinline__ RangeHead range_head(const Geometry &g, const int32_t *__restrict__ tile0,
                                                const uint32_t *__restrict__ tile_ptr, int R, int nranges, int ranges_per_slab)
{
    if (R >= nranges)
        return RangeHead{g.tail_start < g.m ? g.tail_start : -1, (long long)(g.p - 1) * g.tile_elems};
    const int k = R / ranges_per_slab, rho = R % ranges_per_slab;
    const int t0 = tile0[k], n = tile0[k + 1] - t0;
    const int tb = t0 + hot_range_begin(n, rho);
    if (hot_range_begin(n, rho + 1) - hot_range_begin(n, rho) <= 0)
        return RangeHead{-1, 0};
    return RangeHead{(int)(tile_ptr[tb] & ROW_MASK), (long long)tb * g.tile_elems};
}

// conversion time, ONE workgroup: head[R] = first row of range R | RANGE_EXACT, for R = 0 .. nranges (the tail); a range
// without tiles takes the word of the next range that has some (its lead is 0: it joins that range's run, or leads it)
__global__ void __launch_bounds__(1024)
k_range_heads(Geometry g, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ tile0,
              const uint32_t *__restrict__ tile_ptr, int nranges, int ranges_per_slab, uint32_t *__restrict__ head)
{
    for (int R = threadIdx.x; R <= nranges; R += 1024) {
        const RangeHead h = range_head(g, tile0, tile_ptr, R, nranges, ranges_per_slab);
        head[R] = h.row < 0 ? 0xFFFFFFFFu : ((uint32_t)h.row | ((long long)row_ptr[h.row] == h.first ? RANGE_EXACT : 0u));
    }
    __threadfence_block();
    __syncthreads();
    // ranges without tiles (0xFFFFFFFF), from the top down, 1024 at a time: everything above the chunk is final, and a word
    // of the chunk that a neighbour is filling at this moment is either still empty (skipped) or already what this thread
    // is looking for
    for (int base = (nranges / 1024) * 1024; base >= 0; base -= 1024) {
        const int R = base + (int)threadIdx.x;
        if (R <= nranges && head[R] == 0xFFFFFFFFu) {
            uint32_t w = 0xFFFFFFFFu;
            for (int R2 = R + 1; w == 0xFFFFFFFFu && R2 <= nranges; R2++)
                w = __hip_atomic_load(&head[R2], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (w != 0xFFFFFFFFu)
                __hip_atomic_store(&head[R], w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        __threadfence_block();
        __syncthreads();
    }
    for (int R = threadIdx.x; R <= nranges; R += 1024) // (only the ranges behind the last tile are still empty)
        if (head[R] == 0xFFFFFFFFu)
            head[R] = RANGE_NONE;
}

template <typename VT>
__global__ void __launch_bounds__(256)
k_range_finish(Geometry g, const int32_t *__restrict__ row_ptr,
               const VT *__restrict__ val, const VT *__restrict__ xtail, const uint32_t *__restrict__ head,
               VT *__restrict__ P, const VT *__restrict__ lead, int nranges)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    VT *sprod = reinterpret_cast<VT *>(smem); // [T] tail products
    __shared__ VT tail_lead;
    const int tid = threadIdx.x, lane = tid & (OMEGA - 1);
    const int T = g.tile_elems;
    const long long first_tail = (long long)(g.p - 1) * T;
    const int E = (int)((long long)g.nnz - first_tail);
    const int R = blockIdx.x * 256 + tid;
    // this thread's range and its neighbours: their loads go out together with the tail's
    const bool in = R <= nranges;
    const uint32_t hme = in ? head[R] : RANGE_NONE;
    const uint32_t hprev = in && R > 0 ? head[R - 1] : 0xFFFFFFFFu;
    const uint32_t hnext = in && R < nranges ? head[R + 1] : 0xFFFFFFFFu;
    for (int e = tid; e < E; e += 256) // (xtail: the tail's x entries in element order, behind the cold region of the permuted copy)
        sprod[e] = val[first_tail + e] * xtail[e];
    __syncthreads();
    if (g.tail_start < g.m) {
        // first tail row (it may be all of the tail): strided partial sums, then a fixed-shape reduction -- the other rows
        // only in workgroup 0
        {
            __shared__ VT part[4];
            const int b = (int)((long long)row_ptr[g.tail_start + 1] - first_tail);
            VT s0 = 0;
            for (int k = tid; k < b; k += 256)
                s0 += sprod[k];
            s0 = wave_sum(s0);
            if ((tid & (OMEGA - 1)) == 0)
                part[tid >> 6] = s0;
            __syncthreads();
            if (tid == 0)
                tail_lead = (part[0] + part[1]) + (part[2] + part[3]);
        }
        if (blockIdx.x == 0)
            for (int r0 = g.tail_start + 1; r0 < g.m; r0 += 256) { // (uniform trip count: the wavefront works together below)
                const int r = r0 + tid;
                const bool valid = r < g.m;
                int a = 0, b = 0;
                if (valid)
                    a = (int)((long long)row_ptr[r] - first_tail), b = (int)((long long)row_ptr[r + 1] - first_tail);
                // A tail row longer than 32 elements is summed by its whole wavefront (lane-strided partial sums, fixed-shape
                // reduction), as csr5_carry.h tail_rows does: ONE thread walking a hub row's few hundred products through
                // dependent LDS reads was 7 of this kernel's 11.7 us on a row block of hub rows (round 6, timing-only
                // ablations: profiles/r06_probes.txt section 5).
                const bool longrow = valid && b - a > 32;
                VT s1 = 0;
                if (valid && !longrow)
                    for (int k = a; k < b; k++)
                        s1 += sprod[k];
                unsigned long long todo = __ballot(longrow);
                while (todo) {
                    const int src = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const int aa = __shfl(a, src, OMEGA), bb = __shfl(b, src, OMEGA);
                    VT part = 0;
                    for (int k = aa + lane; k < bb; k += OMEGA)
                        part += sprod[k];
                    part = wave_sum(part);
                    if (lane == src)
                        s1 = part;
                }
                if (valid)
                    P[r] = s1;
            }
    }
    __syncthreads();
    const uint32_t row = hme & ROW_MASK;
    // first range of its row?  (rows never decrease along the ranges)
    const bool leader = in && row != RANGE_NONE && (R == 0 || (hprev & ROW_MASK) != row);
    auto lead_of = [&](int R2) -> VT { return R2 == nranges ? tail_lead : lead[R2]; };
    int stop = R + 1; // one past the last range of the run
    if (leader && R < nranges && (hnext & ROW_MASK) == row) {
        int lo = R + 2, hi = nranges + 1; // first index whose row differs, in (R + 1, nranges + 1]
        while (lo < hi) {
            const int mid = (int)(((unsigned)lo + (unsigned)hi) >> 1);
            if ((head[mid] & ROW_MASK) == row)
                lo = mid + 1;
            else
                hi = mid;
        }
        stop = lo;
    }
    const int len = stop - R;
    // the row begins exactly with this range (nobody stored a partial for it) or inside an earlier one
    VT sum = 0;
    if (leader)
        sum = (hme & RANGE_EXACT) ? (VT)0 : P[row];
    if (leader && len <= RUN_WAVE) {
        int R2 = R;
        for (; R2 + 8 <= stop; R2 += 8) { // range order, eight independent loads at a time
            VT part[8];
#pragma unroll
            for (int j = 0; j < 8; j++)
                part[j] = lead_of(R2 + j);
#pragma unroll
            for (int j = 0; j < 8; j++)
                sum += part[j];
        }
        for (; R2 < stop; R2++)
            sum += lead_of(R2);
        P[row] = sum;
    }
    // long runs: the whole wavefront adds the leads (lane-strided partial sums, then a fixed-shape reduction)
    unsigned long long todo = __ballot(leader && len > RUN_WAVE);
    while (todo) {
        const int who = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int r0 = __shfl(R, who, OMEGA), r1 = __shfl(stop, who, OMEGA);
        VT part = 0;
        for (int R2 = r0 + lane; R2 < r1; R2 += OMEGA)
            part += lead_of(R2);
        part = wave_sum(part);
        if (lane == who)
            P[row] = sum + part;
    }
}

// ---- the permuted copy of x ---------------------------------------------------------------------------------------------
// One gather of the entries the hot child reads, in the order it reads them: the table images of all slabs (hot_cols),
// then every slab's cold columns by descending use (cold_cols).  The column lists are read coalesced, x is gathered
// (inside a slab's cold region ties keep column order, so the gather sweeps x upwards class by class), the copy is
// written coalesced.
template <typename VT>
__global__ void __launch_bounds__(256)
k_x_permute(int hot_entries, int cold_total, int tail_entries, const int32_t *__restrict__ hot_cols,
            const int32_t *__restrict__ cold_cols, const int32_t *__restrict__ tail_cols, const VT *__restrict__ x, VT *__restrict__ xp)
{
    constexpr int PER = 4;
    const long long hc = (long long)hot_entries + cold_total, total = hc + tail_entries;
    const long long i0 = ((long long)blockIdx.x * PER) * 256 + threadIdx.x;
    int32_t c[PER];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const long long i = i0 + u * 256;
        c[u] = i < hot_entries ? hot_cols[i] : (i < hc ? cold_cols[i - hot_entries] : (i < total ? tail_cols[i - hc] : 0));
    }
    VT v[PER];
#pragma unroll
    for (int u = 0; u < PER; u++)
        v[u] = x[(uint32_t)c[u]];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const long long i = i0 + u * 256;
        if (i < total)
            xp[i] = v[u];
    }
}

// (the x entries of the child's CSR tail -- at most one tile of them, in the tail's element order -- sit behind the cold
// region: k_range_finish reads them there, so that NO kernel of the hot path reads the caller's vector after this copy)
static long long tail_first(const Geometry &g) { return (long long)(g.p - 1) * g.tile_elems; }
hipError_t launch_x_permute(const Geometry &g, const DeviceArrays &d, int value_type, const void *x, hipStream_t s)
{
    const long long hot_entries = (long long)d.hot_slabs * d.hot_capacity;
    const int tail_entries = g.p > 0 ? (int)((long long)g.nnz - tail_first(g)) : 0;
    const long long total = hot_entries + d.cold_total + tail_entries;
    if (total <= 0)
        return hipSuccess;
    const int32_t *tail_cols = d.col + (g.p > 0 ? tail_first(g) : 0);
    const unsigned blocks = (unsigned)((total + 1023) / 1024);
    if (value_type == CSR5HIP_F64)
        hipLaunchKernelGGL(k_x_permute<double>, dim3(blocks), dim3(256), 0, s, (int)hot_entries, d.cold_total, tail_entries, d.hot_cols,
                           d.cold_cols, tail_cols, (const double *)x, (double *)const_cast<void *>(d.xperm));
    else
        hipLaunchKernelGGL(k_x_permute<float>, dim3(blocks), dim3(256), 0, s, (int)hot_entries, d.cold_total, tail_entries, d.hot_cols,
                           d.cold_cols, tail_cols, (const float *)x, (float *)const_cast<void *>(d.xperm));
    return hipGetLastError();
}

// ---- CSR5HIP_OPT_NARROW_VALUES: an fp64 value stream kept as fp32 when that loses nothing ----------------------------------
// *flag |= 1 if some value is not exactly representable as a NORMAL fp32 number (or +-0, +-inf); NaNs fail the test too
__global__ void __launch_bounds__(256) k_fp32_exact(const double *__restrict__ v, size_t n, unsigned *__restrict__ flag)
{
    bool bad = false;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const double a = v[i];
        const float f = (float)a;
        const double mag = a < 0 ? -a : a;
        bad |= !((double)f == a) || (mag != 0.0 && mag < 1.17549435082228750797e-38);
    }
    if (__ballot(bad) != 0ull && (threadIdx.x & (OMEGA - 1)) == 0)
        *flag = 1u;
}
// o = v as fp32.  v is the hot child's value array: tiles 0 .. tiles-1 of T = 64 sigma elements in lane-major pieces of TWO
// doubles (k_transpose_values), then the CSR tail in CSR order; o gets the same elements in lane-major pieces of FOUR floats
// (16 bytes again: the range kernel's load width), the tail unchanged.
__global__ void __launch_bounds__(256) k_narrow(const double *__restrict__ v, size_t n, float *__restrict__ o, int T, size_t tiles)
{
    const size_t body = tiles * (size_t)T;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        size_t dst = i;
        if (i < body) {
            const size_t t = i / (size_t)T;
            const int r = (int)(i % (size_t)T);
            const int q = r / (2 * OMEGA), l = (r % (2 * OMEGA)) / 2, e = 2 * q + (r & 1); // element e of lane l
            dst = t * (size_t)T + (size_t)((e / 4) * OMEGA + l) * 4 + (e & 3);
        }
        o[dst] = (float)v[i];
    }
}
hipError_t launch_fp32_exact(const double *v, size_t n, unsigned *flag, hipStream_t s)
{
    hipError_t e = hipMemsetAsync(flag, 0, sizeof(unsigned), s);
    if (e != hipSuccess || n == 0)
        return e;
    const size_t want = (n + 255) / 256;
    hipLaunchKernelGGL(k_fp32_exact, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, s, v, n, flag);
    return hipGetLastError();
}
hipError_t launch_narrow(const double *v, size_t n, float *o, int tile_elems, int transposed_tiles, hipStream_t s)
{
    if (n == 0)
        return hipSuccess;
    if (tile_elems % (4 * OMEGA) != 0)
        return hipErrorInvalidValue;
    const size_t want = (n + 255) / 256;
    hipLaunchKernelGGL(k_narrow, dim3((unsigned)(want < 8192 ? want : 8192)), dim3(256), 0, s, v, n, o, tile_elems,
                       (size_t)(transposed_tiles > 0 ? transposed_tiles : 0));
    return hipGetLastError();
}

// ---- dispatch ----------------------------------------------------------------------------------------------------------
// prepare_only: raise the LDS limit of the instantiation this child will launch (160 KB of dynamic LDS) and return.  Done
// once per conversion (csr5_capi.hip build_slabs), not per SpMV and not cached in a process-wide table: the attribute belongs to
// the (device, kernel) pair the handle was converted on, and no two host threads ever race on shared state for it.
template <typename VT, int SIGMA, bool NT, typename ST = VT>
static hipError_t launch_range(const Geometry &g, const DeviceArrays &d, const void *x, void *y, hipStream_t s, bool prepare_only)
{
    HotParams hp{d.hot_slabs, d.hot_slabs / NUM_XCD, d.hot_capacity, d.hot_count, d.hot_tile0, d.col_lo,
                 d.col_hi,    d.slab_off,             d.xperm,        d.cold_base, d.cold_total};
    const size_t lds = (size_t)d.hot_capacity * sizeof(VT) + (size_t)HOT_WAVES * HOT_WAVE_LDS;
    constexpr int DEPTH = CSR5_HOT_DEPTH;
    if constexpr (std::is_same<VT, double>::value && std::is_same<ST, double>::value) {
        if (d.val32) // the child's values are kept as fp32 (every one of them exactly): the instantiation that streams 4-byte values
            return launch_range<VT, SIGMA, NT, float>(g, d, x, y, s, prepare_only);
    }
    auto kern = k_spmv_range<VT, SIGMA, NT, DEPTH, ST>;
    const ST *val = std::is_same<ST, VT>::value ? (const ST *)d.val : (const ST *)d.val32;
    if (prepare_only)
        return hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                   160 * 1024); // (the CU's whole LDS: the same value from every handle, whatever its table size)
    hipError_t e = hipSuccess;
    if (g.p > 1) {
        hipLaunchKernelGGL(kern, dim3(NUM_XCD * HOT_WGS_PER_XCD), dim3(HOT_BLOCK), lds, s, g, val, d.tile_ptr, (VT *)y,
                           (VT *)d.range_lead, hp);
        e = hipGetLastError();
        if (e != hipSuccess)
            return e;
    }
    const int nranges = d.hot_slabs * HOT_RANGES_PER_SLAB;
    const int blocks = (nranges + 1 + 255) / 256;
    const VT *xtail = (const VT *)d.xperm + (size_t)d.hot_slabs * d.hot_capacity + d.cold_total;
    hipLaunchKernelGGL(k_range_finish<VT>, dim3(blocks), dim3(256), (size_t)g.tile_elems * sizeof(VT), s, g, d.row_ptr,
                       (const VT *)d.val, xtail, d.range_head, (VT *)y, (const VT *)d.range_lead, nranges);
    return hipGetLastError();
}

// conversion time (after the child's tile_ptr exists): which row every wavefront range starts in
hipError_t launch_range_heads(const Geometry &g, const DeviceArrays &d, hipStream_t s)
{
    if (g.p <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_range_heads, dim3(1), dim3(1024), 0, s, g, d.row_ptr, d.hot_tile0, d.tile_ptr,
                       d.hot_slabs * HOT_RANGES_PER_SLAB, HOT_RANGES_PER_SLAB, d.range_head);
    return hipGetLastError();
}

// a hot child is converted at sigma = hot_child_sigma(): 8 for fp64, 16 for fp32 (whole dwords of column codes per lane,
// a tile's segments fit the y-compaction region)
template <typename VT>
static hipError_t launch_range_sigma(const Geometry &g, const DeviceArrays &d, const void *x, void *y, bool nt, hipStream_t s,
                                     bool prepare_only)
{
    if (!d.col_lo || !d.xperm)
        return hipErrorInvalidValue;
    switch (g.sigma) {
#define CSR5_HOT_CASE(S)                                                                                               \
    case S:                                                                                                            \
        if constexpr ((size_t)OMEGA * S * sizeof(VT) <= (size_t)HOT_WAVE_LDS)                                          \
            return nt ? launch_range<VT, S, true>(g, d, x, y, s, prepare_only)                                         \
                      : launch_range<VT, S, false>(g, d, x, y, s, prepare_only);                                       \
        else                                                                                                           \
            return hipErrorInvalidValue;
        CSR5_HOT_CASE(8) CSR5_HOT_CASE(16)
#undef CSR5_HOT_CASE
    default: return hipErrorInvalidValue;
    }
}

// the slab child's SpMV when its column words are hot-encoded: P = A' x (y = the partial-sum array of the parent)
hipError_t launch_spmv_hot(const Geometry &g, const DeviceArrays &d, int value_type, const void *x, void *y,
                           const SpmvOptions &opt, hipStream_t s)
{
    if (g.p <= 0)
        return hipSuccess;
    return value_type == CSR5HIP_F64 ? launch_range_sigma<double>(g, d, x, y, opt.stream_nt != 0, s, false)
                                     : launch_range_sigma<float>(g, d, x, y, opt.stream_nt != 0, s, false);
}

// conversion time (and whenever the child's stream policy changes): the LDS limit of the range kernel this child launches
hipError_t prepare_spmv_hot(const Geometry &g, const DeviceArrays &d, int value_type, const SpmvOptions &opt)
{
    if (g.p <= 0)
        return hipSuccess;
    return value_type == CSR5HIP_F64 ? launch_range_sigma<double>(g, d, nullptr, nullptr, opt.stream_nt != 0, nullptr, true)
                                     : launch_range_sigma<float>(g, d, nullptr, nullptr, opt.stream_nt != 0, nullptr, true);
}

} // namespace csr5

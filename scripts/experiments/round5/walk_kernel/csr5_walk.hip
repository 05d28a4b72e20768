// csr5_walk.hip -- the range-walking, software-pipelined tile kernel on the PLAIN CSR5 format arrays (round 5).
//
// What it computes is the reference's tile kernel (CSR5_cuda/detail/cuda/csr5_spmv_cuda.h:59-311: fast / normal track, lane-local
// flag walk, cross-lane segmented sum, empty-row offsets :140-160) plus calibrate (:313-382) and the CSR tail (:384-419) on the
// format arrays of the reference (tile_ptr, tile_desc, offset_pointer, offset, tile-transposed column_index / value), unchanged.
// How the work is laid out is ours:
//   * ONE WAVEFRONT OWNS A CONTIGUOUS RANGE OF TILES (walk_ranges ranges, dealt so that every XCD walks one contiguous part of
//     the matrix).  The row that is open at a tile boundary meets its continuation in the registers of the same wavefront: no
//     per-tile header, no re-read of the successor's first elements, no carry slot and no atomic per tile.  What is left of the
//     reference's calibrate pass is one leading partial per RANGE; it resolves inside the same launch through the arrival protocol
//     of csr5_carry.h (parties = ranges instead of tiles): deterministic, nobody waits, y need not be zeroed.
//   * SOFTWARE PIPELINE, two register sets: tile t+1's column / value streams, its descriptor words, its tile_ptr / offset_ptr
//     pair (scalar cache) and -- x-window variant -- the slice of x it gathers from are requested BEFORE tile t's gathers go out,
//     so a tile costs one memory round trip (streams of t+1 and gathers of t in flight together) instead of two dependent ones.
//   * x-window variant: a wavefront that walks a range can afford a LARGE slice of x in LDS (16 KB: 4 096 fp32 / 2 048 fp64
//     columns, four times the one-tile kernel's) because it restages rarely: the window base comes from a dense per-tile array
//     (scalar load, two tiles ahead), k_tile_tables quantises the bases, and the staged slice stays while consecutive tiles ask
//     for the same base.  A tile whose columns ALL lie inside the slice issues no global gather at all.
//   * a tile's finished rows are compacted in LDS and leave as coalesced stores; tiles with empty rows scatter them through
//     offset[] (requested together with the gathers, so the stores do not wait for a further round trip).
#include "csr5_carry.h"

#ifndef CSR5_WALK_DEPTH
#define CSR5_WALK_DEPTH 3 // tiles whose streams are requested ahead of the tile that computes, + 1 (2 or 3 register sets)
#endif

namespace csr5 {

constexpr uint32_t WALK_EXACT = 0x80000000u; // walk_row bit: the range's first row BEGINS with the range's first element

// first tile of range R when `tiles` tiles are dealt to `nranges` ranges (R == nranges: one past the last tile)
__host__ __device__ __forceinline__ int walk_tile_begin(int R, int nranges, int tiles)
{
    const int q = tiles / nranges, rem = tiles % nranges;
    return R * q + (R < rem ? R : rem);
}

// ---- conversion time: what the ranges' arrival protocol needs (ONE workgroup; nranges <= WALK_MAX_RANGES) ------------------
// walk_row[R]  = first row of range R | WALK_EXACT, R = 0 .. nranges (nranges = the CSR tail as a pseudo-range)
// walk_meta[R] = { expected arrivals | bit 26 long run | bit 27 a closing partial arrives too, head of R's run, 0, 0 }
//                in the layout of the tile-level carry_meta (csr5_format.hip tile_carry_meta), so that carry_arrive / sum_run /
//                k_calibrate work on ranges as they do on tiles.  A run = the consecutive ranges that begin inside the same row.
#if !defined(CSR5_WALK_ONLY_F32) // (the non-template parts live in the f64 half of the two-part build)
__global__ void __launch_bounds__(1024)
k_walk_tables(Geometry g, const int32_t *__restrict__ row_ptr, const uint32_t *__restrict__ tile_ptr, int nranges,
              uint32_t *walk_row, uint4 *__restrict__ walk_meta, uint32_t *__restrict__ long_runs_out)
{
    __shared__ unsigned nlong;
    const int tiles = g.p - 1;
    if (threadIdx.x == 0)
        nlong = 0;
    for (int R = threadIdx.x; R <= nranges; R += 1024) {
        const int tb = R < nranges ? walk_tile_begin(R, nranges, tiles) : tiles;
        const uint32_t row = tile_ptr[tb] & ROW_MASK;
        const bool exact = (long long)row_ptr[row] == (long long)tb * g.tile_elems;
        __hip_atomic_store(&walk_row[R], row | (exact ? WALK_EXACT : 0u), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    __threadfence_block();
    __syncthreads();
    auto row_of = [&](int R) -> uint32_t {
        return __hip_atomic_load(&walk_row[R], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & ROW_MASK;
    };
    for (int R = threadIdx.x; R <= nranges; R += 1024) {
        const uint32_t r = row_of(R);
        int lo = 0, hi = R; // first range whose row is r (rows never decrease along the ranges)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (row_of(mid) < r)
                lo = mid + 1;
            else
                hi = mid;
        }
        uint4 meta = make_uint4(0u, (unsigned)lo, 0u, 0u);
        if (lo == R) {
            lo = R + 1, hi = nranges + 1; // one past the last range of the run
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                if (row_of(mid) <= r)
                    lo = mid + 1;
                else
                    hi = mid;
            }
            const int len = lo - R;
            const bool has_first =
                !(__hip_atomic_load(&walk_row[R], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) & WALK_EXACT);
            meta.x = (unsigned)(len + (has_first ? 1 : 0));
            if (has_first)
                meta.x |= 1u << 27;
            if (len > RUN_SERIAL_MAX) {
                meta.x |= 1u << 26;
                atomicAdd(&nlong, 1u);
            }
        }
        walk_meta[R] = meta;
    }
    __syncthreads();
    if (threadIdx.x == 0 && long_runs_out)
        *long_runs_out = nlong;
}

hipError_t launch_walk_tables(const Geometry &g, const DeviceArrays &d, uint32_t *long_runs_out, hipStream_t s)
{
    if (g.p <= 1 || d.walk_ranges <= 0)
        return hipSuccess;
    hipLaunchKernelGGL(k_walk_tables, dim3(1), dim3(1024), 0, s, g, d.row_ptr, d.tile_ptr, d.walk_ranges, d.walk_row,
                       reinterpret_cast<uint4 *>(d.walk_meta), long_runs_out);
    return hipGetLastError();
}
#endif

// ---- the kernel -----------------------------------------------------------------------------------------------------------
template <typename VT, int SIGMA, bool C16>
struct WalkTile {
    // column words: one per element, or (C16, narrow column codes of k_col16) two 16-bit codes per word + the tile's base
    int32_t c[C16 ? SIGMA / 2 : SIGMA];
    int32_t cbase; // (wave-uniform, scalar cache)
    VT v[SIGMA];
    __device__ __forceinline__ int32_t column(int i) const
    {
        if constexpr (C16)
            return cbase + (int32_t)(((uint32_t)c[i >> 1] >> (16 * (i & 1))) & 0x7FFFu); // (bit 15: row-start flag, unused here)
        else
            return c[i];
    }
    uint32_t w0;         // descriptor word of this lane: y_offset | scansum_offset | bit flags (one packet: sigma <= 16)
    uint32_t tp0, tp1;   // tile_ptr[t], tile_ptr[t + 1] (wave-uniform, scalar cache)
    int32_t offp, offn;  // offset_pointer[t], [t + 1]
};

struct WalkParams {
    const uint32_t *row;  // walk_row
    const uint4 *meta;    // walk_meta
    void *lead, *acc;     // [nranges + 1] of vT: parked leading partials (the calibrator's role) / closing partials
    uint32_t *cnt;        // [nranges + 1] arrival counters
    const int32_t *xwin;  // [p] window base of every tile, -1 = none
    const uint32_t *col16; // narrow column codes (C16 kernels), base16 behind them
    const int32_t *base16;
    int nranges;
};

template <typename VT, int SIGMA, bool XWIN>
constexpr int walk_lds_bytes()
{
    return OMEGA * SIGMA * (int)sizeof(VT) + (XWIN ? WALK_XWIN_BYTES : 0);
}

template <typename VT, int SIGMA, bool XWIN, bool NT, int DEPTH, bool C16 = false>
__global__ void __launch_bounds__(OMEGA)
k_spmv_walk(Geometry g, const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col, const VT *__restrict__ val,
            const VT *__restrict__ x, const uint32_t *__restrict__ tile_ptr, const uint32_t *__restrict__ tile_desc,
            const int32_t *__restrict__ offset_ptr, const int32_t *__restrict__ offset, VT *__restrict__ y, WalkParams wp)
{
    static_assert(num_packet_of(SIGMA) == 1, "one descriptor word per lane");
    static_assert(!C16 || (col16_sigma(SIGMA) && !NT), "narrow column codes: the sigmas k_col16 serves");
    using Tile = WalkTile<VT, SIGMA, C16>;
    using word_t = typename std::conditional<sizeof(VT) == 8, unsigned long long, unsigned>::type;
    constexpr int T = OMEGA * SIGMA;
    constexpr int BIT_Y = bit_y_of(SIGMA), BIT_ALL = BIT_Y + BIT_SS;
    // bytes of the staged slice of x.  Lanes outside it read +0.0 from the LAST slot of the y-compaction region right in front
    // of the slice (never used by a tile: a tile stores at most T - 2 segments there), so that eight wavefronts of the fp32
    // sigma = 16 kernel fill a CU's 160 KB exactly
    constexpr unsigned WXB = WALK_XWIN_BYTES;
    constexpr unsigned ZERO_SLOT = 0u - (unsigned)sizeof(VT);
    extern __shared__ __attribute__((aligned(16))) char smem[];
    VT *const lead = static_cast<VT *>(wp.lead), *const acc = static_cast<VT *>(wp.acc);

    if ((int)blockIdx.x >= wp.nranges) {
        // the CSR tail (csr5_spmv_cuda.h:384-419): extra workgroups of the same grid; its first row's partial is the lead of
        // the pseudo-range behind the last range
        tail_rows<VT, SIGMA>(g, row_ptr, col, val, x, y, (int)blockIdx.x - wp.nranges, reinterpret_cast<VT *>(smem), [&](VT sum) {
            const int R = wp.nranges;
            const uint4 mt = wp.meta[R];
            carry_arrive(acc, wp.cnt, lead, wp.row, (int)mt.y, (int)mt.y == R ? mt.x : wp.meta[mt.y].x, R, false, sum, y);
        });
        return;
    }
    const int lane = threadIdx.x;
    // workgroup b runs on XCD b % 8 (dispatch order; locality only): every XCD walks one contiguous part of the matrix
    int R;
    {
        const int per = wp.nranges / NUM_XCD, rem = wp.nranges % NUM_XCD, xcd = (int)blockIdx.x % NUM_XCD;
        R = xcd * per + (xcd < rem ? xcd : rem) + (int)blockIdx.x / NUM_XCD;
    }
    R = __builtin_amdgcn_readfirstlane(R);
    const int tb = __builtin_amdgcn_readfirstlane(walk_tile_begin(R, wp.nranges, g.p - 1));
    const int te = __builtin_amdgcn_readfirstlane(walk_tile_begin(R + 1, wp.nranges, g.p - 1));

    // wave-uniform tables through the scalar cache (constant address space -> s_load): off the vector memory path
    const auto *tpc = (const __attribute__((address_space(4))) uint32_t *)(uintptr_t)tile_ptr;
    const auto *opc = (const __attribute__((address_space(4))) int32_t *)(uintptr_t)offset_ptr;
    const auto *xwc = (const __attribute__((address_space(4))) int32_t *)(uintptr_t)wp.xwin;
    const auto *b16c = (const __attribute__((address_space(4))) int32_t *)(uintptr_t)wp.base16;
    // the protocol words of this range, requested now and read when the range is done
    const auto *rowc = (const __attribute__((address_space(4))) uint32_t *)(uintptr_t)wp.row;
    const auto *metac = (const __attribute__((address_space(4))) uint32_t *)(uintptr_t)wp.meta; // uint4 per range
    const uint32_t my_meta_x = metac[4 * R], my_meta_y = metac[4 * R + 1];
    const uint32_t next_row = rowc[R + 1];
    const uint32_t next_meta_x = metac[4 * (R + 1)];

    auto *seg = (__attribute__((address_space(3))) VT *)(smem);
    auto *win = (__attribute__((address_space(3))) char *)(smem + (size_t)T * sizeof(VT)); // WXB bytes of x
    const auto xbuf = __builtin_amdgcn_make_buffer_rsrc(const_cast<VT *>(x), (short)0, g.n * (int)sizeof(VT), 0x00020000);

    auto load = [&](Tile &tr, int t) {
        const size_t base = (size_t)t * T + lane;
        const int32_t *ct = col + base;
        const VT *vt = val + base;
        // column words first: the gathers wait for them only
        if constexpr (C16) {
            // sigma / 8 loads of 16 bytes per lane (+ one of 8 for sigma = 12): the layout k_col16 writes
            const uint32_t *ctile = wp.col16 + (size_t)t * (T / 2);
            constexpr int W = SIGMA / 2, G4 = W / 4;
#pragma unroll
            for (int k = 0; k < G4; k++) {
                const uint4 q = reinterpret_cast<const uint4 *>(ctile)[k * OMEGA + lane];
                tr.c[4 * k] = (int32_t)q.x, tr.c[4 * k + 1] = (int32_t)q.y, tr.c[4 * k + 2] = (int32_t)q.z, tr.c[4 * k + 3] = (int32_t)q.w;
            }
            if constexpr (W % 4 != 0) {
                const uint2 q = reinterpret_cast<const uint2 *>(ctile + G4 * 4 * OMEGA)[lane];
                tr.c[4 * G4] = (int32_t)q.x, tr.c[4 * G4 + 1] = (int32_t)q.y;
            }
            tr.cbase = b16c[t];
        } else {
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                tr.c[i] = NT ? __builtin_nontemporal_load(ct + i * OMEGA) : ct[i * OMEGA];
            tr.cbase = 0;
        }
        tr.w0 = tile_desc[(size_t)t * OMEGA + lane];
        tr.tp0 = tpc[t];
        tr.tp1 = tpc[t + 1];
        tr.offp = opc[t];
        tr.offn = opc[t + 1];
#pragma unroll
        for (int i = 0; i < SIGMA; i++)
            tr.v[i] = NT ? __builtin_nontemporal_load(vt + i * OMEGA) : vt[i * OMEGA];
    };

    // ---- x-window (XWIN): `staged` = first column of the slice of x that sits in LDS.  Restaging is rare (once per range
    //      or so) and blocking: 16-byte pieces, eight per lane in flight, straight into LDS.
    int staged = -1;
    auto restage = [&](int wl) {
        if constexpr (XWIN) {
            if (wl >= 0 && wl != staged) { // (wave-uniform)
                typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
                constexpr int PIECES = (int)WXB / 16, CH = 8;
                // (columns behind the end of x read 0 through the buffer's range check; no column word points there)
                const unsigned first = (unsigned)wl * (unsigned)sizeof(VT) + (unsigned)lane * 16u;
#pragma unroll
                for (int p0 = 0; p0 < PIECES; p0 += CH * OMEGA) {
                    u32x4 piece[CH];
#pragma unroll
                    for (int k = 0; k < CH; k++)
                        piece[k] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(
                                                                 xbuf, first + (unsigned)(p0 + k * OMEGA) * 16u, 0, 0));
#pragma unroll
                    for (int k = 0; k < CH; k++)
                        *(__attribute__((address_space(3))) u32x4 *)(win + (size_t)(p0 + k * OMEGA + lane) * 16) = piece[k];
                }
                staged = wl;
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
            }
        }
    };

    // ---- open row (wave-uniform) ------------------------------------------------------------------------------------------
    int open_row = -1;      // the row whose partial is open at the current tile boundary
    VT open_val = 0;        // its partial so far (same value in every lane)
    bool open_is_lead = true; // the row was open (or began) at the range's first element: its partial is this range's lead
    VT lead_val = 0;
    // the open row is complete
    auto emit_open = [&](VT total) {
        if (open_is_lead)
            lead_val = total;
        else if (lane == 0)
            y[open_row] = total;
    };

    // gathers of tile `tr` (window base wl, -1 = none): in-window lanes will read LDS (compute), the others read x through a
    // range-checked buffer load (in-window lanes carry an out-of-range offset there: "return 0, touch no memory") and a bitwise
    // OR merges the two.  A tile whose columns ALL lie in the window issues no buffer load.  Returns "some lane is outside".
    auto gather = [&](const Tile &tr, int wl, word_t (&xg)[SIGMA], int32_t (&offv)[SIGMA]) -> bool {
        bool some_out = true;
        if constexpr (XWIN) {
            // byte offsets: (c - wl) * sizeof(vT) < WXB  <=>  the column lies in the staged slice
            const unsigned wbase = wl >= 0 ? (unsigned)wl * (unsigned)sizeof(VT) : 0x80000000u + WXB; // (no window: all outside)
            bool out = false;
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                out |= (unsigned)tr.column(i) * (unsigned)sizeof(VT) - wbase >= WXB;
            some_out = __ballot(out) != 0ull;
            if (some_out) {
#pragma unroll
                for (int i = 0; i < SIGMA; i++) {
                    const unsigned boff = (unsigned)tr.column(i) * (unsigned)sizeof(VT);
                    const unsigned off = boff - wbase >= WXB ? boff : 0xFFFFFFFFu;
                    if constexpr (sizeof(VT) == 8)
                        xg[i] = __builtin_bit_cast(word_t, __builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 0));
                    else
                        xg[i] = __builtin_bit_cast(word_t, __builtin_amdgcn_raw_buffer_load_b32(xbuf, off, 0, 0));
                }
            }
        } else {
#pragma unroll
            for (int i = 0; i < SIGMA; i++) {
                const unsigned off = (unsigned)tr.column(i) * (unsigned)sizeof(VT);
                if constexpr (sizeof(VT) == 8)
                    xg[i] = __builtin_bit_cast(word_t, __builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 0));
                else
                    xg[i] = __builtin_bit_cast(word_t, __builtin_amdgcn_raw_buffer_load_b32(xbuf, off, 0, 0));
            }
        }
        // empty-row tiles: the row offsets of the tile's segments (csr5_spmv_cuda.h:140-160), one coalesced load per 64 of them,
        // in the same batch as the gathers
        if (__builtin_amdgcn_readfirstlane(tr.tp0) >> 31) { // (wave-uniform)
            const int cnt = tr.offn - tr.offp;
#pragma unroll
            for (int k = 0; k < SIGMA; k++) {
                offv[k] = 0;
                if (k * OMEGA < cnt) {
                    const int j = k * OMEGA + lane;
                    offv[k] = offset[tr.offp + (j < cnt ? j : cnt - 1)];
                }
            }
        }
        return some_out;
    };

    // ---- one tile whose loads (streams in `tr`, gathers in `xg`) are in flight or done ----------------------------------------
    auto compute = [&](const Tile &tr, int wl, bool some_out, const word_t (&xg)[SIGMA],
                       const int32_t (&offv)[SIGMA]) {
        VT mx[SIGMA];
        if constexpr (XWIN) {
            const unsigned wbase = wl >= 0 ? (unsigned)wl * (unsigned)sizeof(VT) : 0x80000000u + WXB;
            word_t lw[SIGMA];
#pragma unroll
            for (int i = 0; i < SIGMA; i++) {
                const unsigned d = (unsigned)tr.column(i) * (unsigned)sizeof(VT) - wbase;
                lw[i] = *(const __attribute__((address_space(3))) word_t *)(win + (int)(d < WXB ? d : ZERO_SLOT)); // outside: +0.0
            }
            if (some_out) { // (wave-uniform)
#pragma unroll
                for (int i = 0; i < SIGMA; i++)
                    lw[i] |= xg[i];
            }
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                mx[i] = __builtin_bit_cast(VT, lw[i]);
        } else {
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                mx[i] = __builtin_bit_cast(VT, xg[i]);
        }
        const uint32_t tp0 = __builtin_amdgcn_readfirstlane(tr.tp0), tp1 = __builtin_amdgcn_readfirstlane(tr.tp1);
        const int rs = (int)(tp0 & ROW_MASK);
        const uint32_t flags = tr.w0 << BIT_ALL; // element i -> bit 31-i
        int y_off = (int)(tr.w0 >> (32 - BIT_Y));
        const bool f0 = (flags >> 31) | (lane == 0);
        const bool present = f0 | ((flags & 0x7FFFFFFFu) != 0);
        if (open_row < 0)
            open_row = rs; // first tile of the range
        if (rs != open_row) {
            // the tile begins with a new row: the open one ended exactly on the boundary
            emit_open(open_val);
            open_row = rs;
            open_val = 0;
            open_is_lead = false;
        }
        if (tp0 == (tp1 & ROW_MASK)) {
            // fast track: the whole tile lies inside the open row (csr5_spmv_cuda.h:59-90)
            VT s = 0;
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                s = fma_vt(tr.v[i], mx[i], s);
            open_val += wave_sum(s);
            return;
        }
        const bool empty_rows = (bool)(tp0 >> 31);
        const unsigned long long pmask = __ballot(present);
        bool direct = f0 && lane != 0;
        VT sum = tr.v[0] * mx[0];
        VT first_sum = 0;
        // steps at which NO lane starts a row cost one fused multiply-add: the test is scalar, on the OR of the lanes' flags
        // (long rows: a tile of nd24k holds two or three row starts in 1 024 elements)
        const uint32_t any_flags = wave_or(flags);
#pragma unroll
        for (int i = 1; i < SIGMA; i++) {
            if ((any_flags >> (31 - i)) & 1u) {
                if ((flags >> (31 - i)) & 1u) {
                    if (direct)
                        seg[y_off] = sum;
                    else
                        first_sum = sum;
                    y_off += direct;
                    direct = true;
                    sum = 0;
                }
            }
            sum = fma_vt(tr.v[i], mx[i], sum);
        }
        if (!direct)
            first_sum = sum;
        // cross-lane step: backward segmented scan R[j] = lead[j] + (present[j] ? 0 : R[j+1]) on DPP row shifts and
        // v_readlane row carries; steps no lane needs are skipped by scalar tests on the flag-owner mask
        VT Rv = f0 ? (VT)0 : first_sum;
        const unsigned long long z1 = ~pmask;
        if (z1) {
            const unsigned long long ahead = pmask >> lane;
            const int dist = ahead ? __builtin_ctzll(ahead) : OMEGA - 1 - lane;
            {
                const VT up = dpp_move<DPP_ROW_SHL1>(Rv);
                Rv += dist >= 1 ? up : (VT)0;
            }
            const unsigned long long z2 = z1 & (z1 >> 1);
            if (z2) {
                {
                    const VT up = dpp_move<DPP_ROW_SHL2>(Rv);
                    Rv += dist >= 2 ? up : (VT)0;
                }
                const unsigned long long z4 = z2 & (z2 >> 2);
                if (z4) {
                    {
                        const VT up = dpp_move<DPP_ROW_SHL4>(Rv);
                        Rv += dist >= 4 ? up : (VT)0;
                    }
                    if (z4 & (z4 >> 4)) {
                        const VT up = dpp_move<DPP_ROW_SHL8>(Rv);
                        Rv += dist >= 8 ? up : (VT)0;
                    }
                }
            }
            const int reach = lane + dist;
#pragma unroll
            for (int edge = 48; edge >= 16; edge -= 16) {
                if (!((pmask >> (edge - 1)) & 1ull)) { // lane edge-1 owns no flag: its run crosses the edge
                    const VT carry_in = bcast_lane(Rv, edge);
                    Rv += ((lane >> 4) == (edge >> 4) - 1 && reach >= edge) ? carry_in : (VT)0;
                }
            }
        }
        const VT S = lane_above(Rv); // lane 63 gets 0
        if (present)
            sum += S;
        // leading run of the tile (elements before the first row start at position >= 1): continues the open row
        const VT leading = bcast_lane(direct ? first_sum : sum, 0);
        const unsigned long long dmask = __ballot(direct);
        if (!dmask) {
            open_val += leading; // no row starts inside the tile
            return;
        }
        // Row rs (the open row) is complete now; slots 0 .. nslot-2 are rows that start and end inside the tile, slot nslot-1
        // -- the last segment of the highest flag-owning lane -- stays open.  Slot j = row rs + 1 + j, or, in a tile with
        // empty rows, rs + 1 + offset[offset_pointer[t] + j].
        const int last = 63 - __builtin_clzll(pmask);
        const int nslot = __builtin_amdgcn_readlane(y_off, last) + 1;
        const VT closing = bcast_lane(sum, last);
        if (direct && lane != last)
            seg[y_off] = sum;
        emit_open(open_val + leading);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        VT *const y_local = y + rs + 1;
        int open_rel = nslot - 1;
        if (empty_rows) {
            const int kk = (nslot - 1) >> 6, ll = (nslot - 1) & 63;
#pragma unroll
            for (int k = 0; k < SIGMA; k++) {
                if (k * OMEGA < nslot - 1) {
                    const int j = k * OMEGA + lane;
                    if (j < nslot - 1)
                        y_local[offv[k]] = seg[j];
                }
                if (k == kk)
                    open_rel = __builtin_amdgcn_readlane(offv[k], ll);
            }
        } else {
            for (int j = lane; j < nslot - 1; j += OMEGA)
                y_local[j] = seg[j];
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront"); // the next tile's LDS writes stay behind these reads
        __builtin_amdgcn_wave_barrier();
        open_row = rs + 1 + open_rel;
        open_val = closing;
        open_is_lead = false;
    };

    // ---- the walk -----------------------------------------------------------------------------------------------------------
    // One step = one tile: this tile's gathers FIRST (vector loads return in order: they must not queue behind the streams), then
    // the streams of the tile DEPTH - 1 ahead, which stay in flight while this tile computes; the tile in between is in flight
    // already.  DEPTH register sets rotate without copies (the loop is unrolled DEPTH times, the last tiles are peeled).  HBM
    // latency under load is ~2.5 us here: with one tile of streams in flight per wavefront (DEPTH 2) and the 8 wavefronts per CU
    // the 16-KB window leaves room for, a CU holds 64 KB in flight and the chip 5 TB/s; DEPTH 3 doubles that.
    word_t xg[SIGMA];
    int32_t offv[SIGMA];
    int wl_cur = -1;
    auto step = [&](Tile &cur, Tile &ahead, int tt, auto load_ahead) {
        int wl_nx = -1;
        if constexpr (XWIN)
            wl_nx = xwc[tt + 1 < te ? tt + 1 : tt]; // the next tile's window: read when this tile is done
        __builtin_amdgcn_sched_barrier(0);
        const bool some_out = gather(cur, wl_cur, xg, offv);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (decltype(load_ahead)::value)
            load(ahead, tt + DEPTH - 1 < te ? tt + DEPTH - 1 : te - 1); // (behind the range's end: its last tile again, unused)
        __builtin_amdgcn_sched_barrier(0);
        compute(cur, wl_cur, some_out, xg, offv);
        restage(wl_nx);
        wl_cur = wl_nx;
        __builtin_amdgcn_sched_barrier(0);
    };
    Tile a, b;
    load(a, tb);
    if constexpr (XWIN) {
        wl_cur = xwc[tb];
        *(__attribute__((address_space(3))) word_t *)(win + (int)ZERO_SLOT) = 0; // what the out-of-window lanes read
        restage(wl_cur);
    }
    int t = tb;
    if constexpr (DEPTH == 2) {
        for (; t + 2 <= te; t += 2) {
            step(a, b, t, std::true_type{});
            step(b, a, t + 1, std::true_type{});
        }
        if (t < te)
            step(a, a, t, std::false_type{});
    } else {
        Tile c;
        load(b, tb + 1 < te ? tb + 1 : tb);
        for (; t + 3 <= te; t += 3) {
            step(a, c, t, std::true_type{});
            step(b, a, t + 1, std::true_type{});
            step(c, b, t + 2, std::true_type{});
        }
        if (t < te)
            step(a, a, t, std::false_type{});
        if (t + 1 < te)
            step(b, b, t + 1, std::false_type{});
    }

    // ---- the seams of this range: its lead, and the row that is open at its end --------------------------------------------
    if (open_is_lead)
        lead_val = open_val; // the whole range lies inside one row
    if (lane == 0) {
        const int slot = (int)my_meta_y;
        carry_arrive(acc, wp.cnt, lead, wp.row, slot, slot == R ? my_meta_x : wp.meta[slot].x, R, false, lead_val, y);
        if (!open_is_lead) {
            if ((int)(next_row & ROW_MASK) == open_row && !(next_row & WALK_EXACT))
                carry_arrive(acc, wp.cnt, lead, wp.row, R + 1, next_meta_x, R, true, open_val, y);
            else
                y[open_row] = open_val;
        }
    }
}

// ---- dispatch -------------------------------------------------------------------------------------------------------------
template <typename VT, int SIGMA, bool XWIN, bool NT, bool C16 = false>
static hipError_t launch_walk_one(const Geometry &g, const DeviceArrays &d, const void *x, void *y, const SpmvOptions &opt,
                                  hipStream_t s)
{
    const int tail_rows_n = g.m - g.tail_start;
    const int tail_blocks = tail_rows_n > 0 ? (tail_rows_n + BLOCK - 1) / BLOCK : 0;
    WalkParams wp{d.walk_row, reinterpret_cast<const uint4 *>(d.walk_meta), d.walk_lead, d.walk_acc, d.walk_cnt, d.xwin_base,
                  d.col16, d.base16, d.walk_ranges};
    constexpr size_t lds = (size_t)walk_lds_bytes<VT, SIGMA, XWIN>();
    hipLaunchKernelGGL((k_spmv_walk<VT, SIGMA, XWIN, NT, CSR5_WALK_DEPTH, C16>), dim3(d.walk_ranges + tail_blocks), dim3(OMEGA), lds, s, g, d.row_ptr, d.col, (const VT *)d.val, (const VT *)x,
                       d.tile_ptr, d.tile_desc, d.offset_ptr, d.offset, (VT *)y, wp);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess || !opt.walk_long_runs)
        return e;
    // a row that spans more than RUN_SERIAL_MAX ranges: its parties only parked their partials
    return launch_calibrate_long(d.walk_ranges + 1, g.m, sizeof(VT) == 8 ? CSR5HIP_F64 : CSR5HIP_F32, d.walk_row, d.walk_meta,
                                 d.walk_lead, d.walk_acc, y, s);
}

template <typename VT>
static hipError_t launch_walk_sigma(const Geometry &g, const DeviceArrays &d, const void *x, void *y, const SpmvOptions &opt,
                                    hipStream_t s)
{
    switch (g.sigma) {
#define CSR5_WALK_CASE(S)                                                                                                      \
    case S:                                                                                                                    \
        if constexpr (col16_sigma(S))                                                                                          \
            if (opt.walk_x_window && opt.col16 && d.col16)                                                                     \
                return launch_walk_one<VT, S, true, false, true>(g, d, x, y, opt, s);                                          \
        if (opt.walk_x_window)                                                                                                 \
            return opt.stream_nt ? launch_walk_one<VT, S, true, true>(g, d, x, y, opt, s)                                      \
                                 : launch_walk_one<VT, S, true, false>(g, d, x, y, opt, s);                                    \
        return opt.stream_nt ? launch_walk_one<VT, S, false, true>(g, d, x, y, opt, s)                                         \
                             : launch_walk_one<VT, S, false, false>(g, d, x, y, opt, s);
        CSR5_WALK_CASE(4) CSR5_WALK_CASE(5) CSR5_WALK_CASE(6) CSR5_WALK_CASE(7) CSR5_WALK_CASE(8) CSR5_WALK_CASE(9)
        CSR5_WALK_CASE(10) CSR5_WALK_CASE(11) CSR5_WALK_CASE(12) CSR5_WALK_CASE(13) CSR5_WALK_CASE(14) CSR5_WALK_CASE(15)
        CSR5_WALK_CASE(16)
#undef CSR5_WALK_CASE
    default: return hipErrorInvalidValue;
    }
}

#if !defined(CSR5_WALK_ONLY_F32)
// can the walking kernel run this matrix?  (sigma <= 16: one descriptor packet; x within the reach of a 32-bit buffer offset)
bool walk_supported(const Geometry &g, int value_size)
{
    if (g.p <= 1 || g.sigma < 4 || g.sigma > WALK_MAX_SIGMA)
        return false;
    return (long long)g.n * value_size < (1LL << 31);
}
// dynamic LDS of one wavefront (= workgroup) of the walking kernel
int walk_wave_lds_bytes(int sigma, int value_size, int x_window)
{
    return OMEGA * sigma * value_size + (x_window ? WALK_XWIN_BYTES : 0);
}
#endif

#if !defined(CSR5_WALK_ONLY_F32)
hipError_t launch_spmv_walk_f64(const Geometry &g, const DeviceArrays &d, const void *x, void *y, const SpmvOptions &opt,
                                hipStream_t s)
{
    return launch_walk_sigma<double>(g, d, x, y, opt, s);
}
#endif
#if !defined(CSR5_WALK_ONLY_F64)
hipError_t launch_spmv_walk_f32(const Geometry &g, const DeviceArrays &d, const void *x, void *y, const SpmvOptions &opt,
                                hipStream_t s)
{
    return launch_walk_sigma<float>(g, d, x, y, opt, s);
}
#endif

} // namespace csr5

// csr5_carry.h -- the carry protocol between parties that hold partial sums of one row (the one-tile-per-wavefront kernel of
// csr5_spmv.hip: parties = tiles), and the CSR tail tile.  gfx950 only; not a public header.
#pragma once

#include "csr5_internal.h"
#include "csr5_wave.h"

#include <type_traits>

namespace csr5 {

// ---- fused-mode carry protocol ------------------------------------------------------------------
// Slot h (= first tile of a run of tiles that begin inside the same row r) collects every partial of
// row r that is cut by a tile boundary and is not covered by short-spill ownership.  `expected` =
// number of partials that will arrive at the slot (carry_meta[h].x, known at conversion time):
//   1  the partial IS the row (row starts on the tile boundary and ends inside the tile): plain store.
//   2  exchange handshake, ONE returning atomic per party: each party swaps the bit-inverted value
//      into the slot (0 = empty, the memset state); whoever gets a non-zero word back is second, adds
//      the two partials (a+b == b+a: bit-reproducible), stores y and re-arms the slot.  (The
//      all-ones NaN payload, whose inverse would read as "empty", is published as the default quiet NaN.)
//   >2 rows spanning several tiles: every party parks its partial in its OWN word (leading partial of
//      tile t -> calibrator[t], closing partial of tile h-1 -> acc[h]) with a write-through agent-scope
//      store, drains it (s_waitcnt vmcnt(0)), then bumps the arrival counter.  The last arriver reads the
//      words back with agent-scope loads and adds them IN TILE ORDER -- the same association as the
//      two-pass k_calibrate, so the result is bit-reproducible -- stores y and re-arms the counter.
//   long runs (> RUN_SERIAL_MAX tiles): parties only park their partial; k_calibrate finishes them.
// All slot accesses are device-scope atomics: performed at the memory side, coherent across the 8 XCD
// L2s, no dependence on dispatch order or placement, nobody ever waits for another workgroup.
// Sum of the parked partials of the run headed by tile `slot`, in a fixed order shared by the fused
// kernel (ATOMIC loads, same launch) and k_calibrate (plain loads, next launch):
//   len <= RUN_SERIAL_MAX : first + cal[slot] + cal[slot+1] + ...                    (one lane, `leader`)
//   longer                : first + wave_sum(lane-strided partial sums of cal[slot..]) (whole wavefront)
// `first` = the closing partial of tile slot-1 when the row starts inside it (has_first).
// The long form must be called by all 64 lanes with wave-uniform arguments; result valid in `leader`.
template <typename VT, bool ATOMIC>
__device__ __forceinline__ VT sum_run(const VT *calibrator, int slot, int len, bool has_first, VT first,
                                      int lane, int leader)
{
    using bits_t = typename std::conditional<sizeof(VT) == 8, unsigned long long, unsigned>::type;
    auto load = [&](int k) -> VT {
        if constexpr (ATOMIC)
            return __builtin_bit_cast(VT, __hip_atomic_load(reinterpret_cast<const bits_t *>(&calibrator[slot + k]),
                                                            __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
        else
            return calibrator[slot + k];
    };
    VT total = 0;
    if (len <= RUN_SERIAL_MAX) {
        if (lane == leader) {
            total = has_first ? first + load(0) : load(0);
            // same left-to-right association as before, but eight loads are in flight at a time: a one-lane
            // chain of dependent round trips (up to 64 of them) was the whole cost of rows spanning 8-64 tiles
            int k = 1;
            for (; k + 8 <= len; k += 8) {
                VT part[8];
#pragma unroll
                for (int j = 0; j < 8; j++)
                    part[j] = load(k + j);
#pragma unroll
                for (int j = 0; j < 8; j++)
                    total += part[j];
            }
            for (; k < len; k++)
                total += load(k);
        }
    } else {
        VT part = 0;
#pragma unroll 4
        for (int k = lane; k < len; k += OMEGA)
            part += load(k);
        part = wave_sum(part);
        total = has_first ? first + part : part;
    }
    return total;
}

template <typename VT>
__device__ __forceinline__ void carry_arrive(VT *acc, uint32_t *cnt, VT *calibrator,
                                             const uint32_t *tile_ptr, int slot, uint32_t meta_x,
                                             int my_tile, bool is_closing, VT v, VT *y)
{
    using bits_t = typename std::conditional<sizeof(VT) == 8, unsigned long long, unsigned>::type;
    const uint32_t expected = meta_x & 0x00FFFFFFu;
    VT *row_y = y + (tile_ptr[slot] & ROW_MASK);
    if ((meta_x >> 26) & 1u) {
        // long run (> RUN_SERIAL_MAX tiles, e.g. a row with 10^5..10^7 non-zeros): park the partial with a
        // plain store; k_calibrate<.., true> (second launch, only for matrices that have such rows) sums
        // them with a whole wavefront.  No counter: 10^4 arrivals on one word would serialise.
        *(is_closing ? &acc[slot] : &calibrator[my_tile]) = v;
    } else if (expected == 1u) {
        *row_y = v;
    } else if (expected == 2u) {
        bits_t *s = reinterpret_cast<bits_t *>(&acc[slot]);
        // 0 marks the empty slot, so the one payload whose inverse is 0 -- the all-ones NaN that poisoned
        // inputs (0xFF fill) propagate through the FMAs -- is published as the default quiet NaN instead:
        // both parties would otherwise believe they came first and the slot would stay armed for good.
        bits_t vb = __builtin_bit_cast(bits_t, v);
        if (vb == ~(bits_t)0)
            vb = sizeof(VT) == 8 ? (bits_t)0x7FF8000000000000ull : (bits_t)0x7FC00000u;
        const bits_t mine = ~vb;
        const bits_t other = __hip_atomic_exchange(s, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (other != 0) {
            *row_y = v + __builtin_bit_cast(VT, (bits_t)~other);
            __hip_atomic_store(s, (bits_t)0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    } else {
        // park this party's partial in its own word, write-through, and drain it before arriving
        bits_t *mine = reinterpret_cast<bits_t *>(is_closing ? &acc[slot] : &calibrator[my_tile]);
        __hip_atomic_store(mine, __builtin_bit_cast(bits_t, v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const uint32_t arrived =
            __hip_atomic_fetch_add(&cnt[slot], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + 1u;
        if (arrived == expected) {
            const bool has_first = (meta_x >> 27) & 1u;
            const int len = (int)expected - (has_first ? 1 : 0); // <= RUN_SERIAL_MAX here
            VT first = 0;
            if (has_first)
                first = __builtin_bit_cast(VT, __hip_atomic_load(reinterpret_cast<bits_t *>(&acc[slot]),
                                                                 __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            const VT total = sum_run<VT, true>(calibrator, slot, len, has_first, first, 0, 0);
            __hip_atomic_store(&cnt[slot], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *row_y = total;
        }
    }
}

// ---- CSR tail tile: rows tail_start..m-1, non-zeros from (p-1)*T, untransposed -------------------
// The tail holds E <= T <= 2048 non-zeros but may span any number of (mostly empty) rows.  Every tail
// workgroup owns 256 consecutive rows.  Latency shape = a tile's: ONE round trip fetches all E
// column/value pairs and the workgroup's row pointers, a second one gathers x; products go to LDS and
// each thread then sums its own row from LDS (rows longer than 32 are summed by the whole wave).
// The first row's partial is a carry.
constexpr int TAIL_MAX = OMEGA * CSR5HIP_MAX_SIGMA; // 2048

template <typename VT, int SIGMA, typename LeadSink>
__device__ __forceinline__ void tail_rows(const Geometry &g, const int32_t *__restrict__ row_ptr,
                                          const int32_t *__restrict__ col,
                                          const VT *__restrict__ val, const VT *__restrict__ x,
                                          VT *__restrict__ y, int tail_block, VT *sprod, LeadSink lead_sink)
{
    const int tid = threadIdx.x;
    const int lane = tid & (OMEGA - 1);
    const int first_tail = (g.p - 1) * g.tile_elems;
    const int E = g.nnz - first_tail;
    // elements per thread: the tail holds at most T = 64*sigma non-zeros (sized per instantiation so
    // that the tail path does not dictate the register allocation of the small-sigma kernels)
    constexpr int PER = ((SIGMA > 0 ? OMEGA * SIGMA : TAIL_MAX) + BLOCK - 1) / BLOCK;
    int32_t c[PER];
    VT v[PER];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int e = tid + k * BLOCK;
        const int idx = first_tail + (e < E ? e : 0);
        c[k] = col[idx];
        v[k] = val[idx];
    }
    const int r = g.tail_start + tail_block * BLOCK + tid;
    const bool valid = r < g.m;
    int a = 0, b = 0;
    if (valid) {
        a = row_ptr[r];
        b = row_ptr[r + 1];
    }
    // unconditional gathers (out-of-range slots re-read element 0's column): one round trip, no
    // per-gather wait; only the LDS store is predicated
    VT xv[PER];
#pragma unroll
    for (int k = 0; k < PER; k++)
        xv[k] = x[c[k]];
#pragma unroll
    for (int k = 0; k < PER; k++) {
        const int e = tid + k * BLOCK;
        if (e < E)
            sprod[e] = v[k] * xv[k];
    }
    __syncthreads();
    a = (r == g.tail_start ? first_tail : a) - first_tail;
    b -= first_tail;
    const bool longrow = valid && (b - a) > 32;
    VT sum = 0;
    if (valid && !longrow)
        for (int k = a; k < b; k++)
            sum += sprod[k];
    unsigned long long todo = __ballot(longrow);
    while (todo) {
        const int src = __builtin_ctzll(todo);
        todo &= todo - 1;
        const int aa = __shfl(a, src, OMEGA);
        const int bb = __shfl(b, src, OMEGA);
        VT s = 0;
        for (int k = aa + lane; k < bb; k += OMEGA)
            s += sprod[k];
        s = wave_sum(s);
        if (lane == src)
            sum = s;
    }
    if (!valid)
        return;
    if (r == g.tail_start)
        lead_sink(sum); // the first tail row may have begun before the tail: its partial is a carry
    else
        y[r] = sum;
}

} // namespace csr5

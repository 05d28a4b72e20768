// Micro-probe (round 4): does it matter to the L1's miss queue HOW the column / value streams of the persistent kernel arrive?
// Shape of k_spmv_range: 256 workgroups x 512 threads (one per CU, 8 wavefronts), every wavefront walks 512-element tiles of
// a column (4 B) + value (8 B) stream, per tile 8 "x gathers" of which COLD % of the lanes go to memory (a region of XKB
// kilobytes per XCD: L2-resident when small), the others read an LDS table, plus WORK (default 60) dependent DPP + add steps.
//   MODE 0: streams through vector registers, one tile ahead (what the product does)
//   MODE 1: streams through LDS-DMA (global_load_lds_dwordx4: L2 -> LDS without registers), one tile ahead, then LDS -> registers
//   MODE 2: no streams at all (gathers + work only)        MODE 3: streams only (no gathers)
// Prints microseconds per launch.   usage: lds_dma_streams [nnz = 2^28] [cold % = 28] [x KB per XCD = 3600]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#ifndef WORK
#define WORK 60
#endif
constexpr int SIGMA = 8, T = 64 * SIGMA, TABLE = 8160, WAVES = 8, STAGE = T * 12; // 6 KB per tile
// -DINTERLEAVE: ONE array of 6-KB tile records [512 column words | 512 values] instead of two arrays (a wavefront then reads one
// sequential stream instead of two)
#ifdef INTERLEAVE
__host__ __device__ inline const int *col_of(const int *col, const double *, size_t tt) { return reinterpret_cast<const int *>(reinterpret_cast<const char *>(col) + tt * STAGE); }
__host__ __device__ inline const double *val_of(const int *col, const double *, size_t tt) { return reinterpret_cast<const double *>(reinterpret_cast<const char *>(col) + tt * STAGE + T * 4); }
#else
__host__ __device__ inline const int *col_of(const int *col, const double *, size_t tt) { return col + tt * T; }
__host__ __device__ inline const double *val_of(const int *, const double *val, size_t tt) { return val + tt * T; }
#endif

__global__ void k_fill(int *col, double *val, size_t n, int coldpct, int xcols)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const bool cold = (int)(h % 100) < coldpct;
        const unsigned r = (unsigned)(h >> 20);
        const_cast<int *>(col_of(col, val, i / T))[i % T] = cold ? (int)(r % (unsigned)xcols) : (int)(0x80000000u | (1u + r % (TABLE - 1)));
        const_cast<double *>(val_of(col, val, i / T))[i % T] = 1.0;
    }
}

template <int MODE>
__global__ void __launch_bounds__(WAVES * 64) k_probe(const int *__restrict__ col, const double *__restrict__ val,
                                                      const double *__restrict__ x, int xbytes_per_xcd, size_t ntiles,
                                                      double *__restrict__ out, int coldpct, unsigned xcols)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto *hot = (__attribute__((address_space(3))) double *)(smem);
    for (int j = threadIdx.x; j < TABLE; j += WAVES * 64)
        hot[j] = j ? 1.0 : 0.0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, xcd = blockIdx.x % 8;
    char *stage = smem + TABLE * 8 + wave * STAGE;
    const auto xbuf = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x) + (size_t)xcd * (xbytes_per_xcd / 8), (short)0,
                                                         xbytes_per_xcd, 0x00020000);
    const size_t nw = (size_t)gridDim.x * WAVES, w = (size_t)blockIdx.x * WAVES + wave;
    const size_t q = ntiles / nw;
    size_t t = w * q;
    const size_t t1 = t + q;
    double acc = 0;
    int c[SIGMA];
    double v[SIGMA];
    auto load_regs = [&](size_t tt) {
        const int *ct = col_of(col, val, tt) + lane;
        const double *vt = val_of(col, val, tt) + lane;
#pragma unroll
        for (int i = 0; i < SIGMA; i++)
            c[i] = __builtin_nontemporal_load(ct + i * 64);
#ifdef WIDE_VAL // four 16-byte loads per lane instead of eight 8-byte ones (same lines, half the vector-memory instructions)
        typedef double d2 __attribute__((ext_vector_type(2)));
        const d2 *vt2 = reinterpret_cast<const d2 *>(val_of(col, val, tt)) + lane;
#pragma unroll
        for (int i = 0; i < SIGMA / 2; i++) {
            const d2 w = __builtin_nontemporal_load(vt2 + i * 64);
            v[2 * i] = w.x, v[2 * i + 1] = w.y;
        }
#else
#pragma unroll
        for (int i = 0; i < SIGMA; i++)
            v[i] = __builtin_nontemporal_load(vt + i * 64);
#endif
    };
    // LDS-DMA: the tile's 2 KB of column words (2 wave loads of 16 B per lane) and 4 KB of values (4 wave loads)
    auto dma = [&](size_t tt) {
        const char *gc = reinterpret_cast<const char *>(col + tt * T) + lane * 16;
        const char *gv = reinterpret_cast<const char *>(val + tt * T) + lane * 16;
#pragma unroll
        for (int i = 0; i < 2; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gc + i * 1024),
                                             (__attribute__((address_space(3))) void *)(stage + i * 1024), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 4; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gv + i * 1024),
                                             (__attribute__((address_space(3))) void *)(stage + 2048 + i * 1024), 16, 0, 0);
    };
    auto from_stage = [&]() {
        const int *sc = reinterpret_cast<const int *>(stage);
        const double *sv = reinterpret_cast<const double *>(stage + 2048);
#pragma unroll
        for (int i = 0; i < SIGMA; i++)
            c[i] = sc[i * 64 + lane];
#pragma unroll
        for (int i = 0; i < SIGMA; i++)
            v[i] = sv[i * 64 + lane];
    };
    int cn[SIGMA];
    double vn[SIGMA];
    if ((MODE == 0 || MODE == 3) && t < t1)
        load_regs(t);
    if (MODE == 1 && t < t1) {
        dma(t);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    for (; t < t1; t++) {
        if (MODE == 1) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            from_stage();
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        if (MODE == 2) { // no streams: column codes from a cheap 32-bit mix (a few VALU per element), values constant
#pragma unroll
            for (int i = 0; i < SIGMA; i++) {
                unsigned h = ((unsigned)t * (unsigned)T + i * 64 + lane) * 0x9E3779B1u;
                h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;
                const unsigned r = h >> 7;
                c[i] = (h & 127u) < (unsigned)(coldpct * 128 / 100) ? (int)__umulhi(r << 7, xcols) : (int)(0x80000000u | (1u + (r & (TABLE / 2 - 1))));
                v[i] = 1.0;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        unsigned long long g[SIGMA];
#ifdef COMPACT_GATHERS // the tile's cold lanes as FULL gather instructions (what a cross-lane compaction would issue): the same number
                       // of cold lines, ceil(cold / 64) instructions instead of 8 -- timing only, the values go to arbitrary elements
        {
            int ncold = 0;
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                ncold += c[i] >= 0;
#pragma unroll
            for (int d = 32; d >= 1; d >>= 1)
                ncold += __shfl_xor(ncold, d, 64);
#pragma unroll
            for (int i = 0; i < SIGMA; i++) {
                g[i] = 0ull;
                if (i * 64 < ncold && MODE != 3) { // (wave-uniform)
                    unsigned h = ((unsigned)t * 512u + i * 64 + lane) * 0x9E3779B1u; // a fresh pseudo-random entry of the region per lane:
                    h ^= h >> 15; h *= 0x85EBCA77u; h ^= h >> 13;                       // as many distinct lines as the 8-instruction form
                    const unsigned pick = __umulhi(h, (unsigned)(xbytes_per_xcd / 8));
                    const unsigned off = i * 64 + lane < ncold ? pick * 8u : 0xFFFFFFFFu;
                    g[i] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 0));
                }
            }
        }
#else
#pragma unroll
        for (int i = 0; i < SIGMA; i++) {
            const unsigned off = (c[i] < 0 || MODE == 3) ? 0xFFFFFFFFu : (unsigned)c[i] * 8u;
            g[i] = MODE == 3 ? 0ull : __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 0));
        }
#endif
        __builtin_amdgcn_sched_barrier(0);
        const size_t tn = t + 1 < t1 ? t + 1 : t;
        if (MODE == 0 || MODE == 3) { // next tile's streams into the second register set
            const int *ct = col_of(col, val, tn) + lane;
            const double *vt = val_of(col, val, tn) + lane;
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                cn[i] = __builtin_nontemporal_load(ct + i * 64);
#ifdef WIDE_VAL
            typedef double d2 __attribute__((ext_vector_type(2)));
            const d2 *vt2 = reinterpret_cast<const d2 *>(val_of(col, val, tn)) + lane;
#pragma unroll
            for (int i = 0; i < SIGMA / 2; i++) {
                const d2 w = __builtin_nontemporal_load(vt2 + i * 64);
                vn[2 * i] = w.x, vn[2 * i + 1] = w.y;
            }
#else
#pragma unroll
            for (int i = 0; i < SIGMA; i++)
                vn[i] = __builtin_nontemporal_load(vt + i * 64);
#endif
        }
        if (MODE == 1)
            dma(tn);
        __builtin_amdgcn_sched_barrier(0);
        double s = 0;
#pragma unroll
        for (int i = 0; i < SIGMA; i++) {
            const unsigned long long tw = __builtin_bit_cast(unsigned long long, hot[c[i] < 0 ? (unsigned)c[i] & 0x7FFFFFFFu : 0u]);
            s = __builtin_fma(v[i], __builtin_bit_cast(double, g[i] | tw), s);
        }
#pragma unroll
        for (int k = 0; k < WORK; k++) {
            const unsigned long long b = __builtin_bit_cast(unsigned long long, s);
            const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x111, 0xF, 0xF, true);
            const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x111, 0xF, 0xF, true);
            s += __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo) * 1e-30;
        }
        acc += s;
        if (MODE == 0 || MODE == 3) {
#pragma unroll
            for (int i = 0; i < SIGMA; i++) {
                c[i] = cn[i];
                v[i] = vn[i];
            }
        }
        if (MODE == 1)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    if (acc == 0.12345)
        out[w * 64 + lane] = acc;
}

// MODE 4: wavefront specialisation.  CONS consumer wavefronts (gathers + table reads + work) and PROD producer wavefronts that
// only stream: a producer feeds CONS / PROD consumers through two 6-KB LDS buffers each (global_load_lds_dwordx4, flags in LDS).
// Vector-memory results return IN ORDER per wavefront: in MODE 0 a tile's gathers (L2-hit latency) are issued behind the next
// tile's stream loads (HBM latency) or the other way round, and one of the two always waits for the other; here the two classes
// of requests sit in different wavefronts' queues.
constexpr int CONS = 8;
template <int PROD>
__global__ void __launch_bounds__((CONS + PROD) * 64) k_split(const int *__restrict__ col, const double *__restrict__ val,
                                                              const double *__restrict__ x, int xbytes_per_xcd, size_t ntiles,
                                                              double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto *hot = (__attribute__((address_space(3))) double *)(smem);
    auto *flags = (volatile __attribute__((address_space(3))) int *)(smem + TABLE * 8 + CONS * 2 * STAGE);
    for (int j = threadIdx.x; j < TABLE; j += (CONS + PROD) * 64)
        hot[j] = j ? 1.0 : 0.0;
    if (threadIdx.x < CONS * 2)
        flags[threadIdx.x] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, xcd = blockIdx.x % 8;
    const size_t nw = (size_t)gridDim.x * CONS;
    const size_t q = ntiles / nw;
    char *stages = smem + TABLE * 8;
    if (wave >= CONS) {
        constexpr int PER = CONS / PROD, D = PER * 2;
        const int p = wave - CONS;
        const size_t total = (size_t)PER * q;
        auto target = [&](size_t n, int &c, size_t &k) { c = p * PER + (int)(n % PER); k = n / PER; };
        for (size_t n = 0; n < total; n++) {
            int c; size_t k;
            target(n, c, k);
            const int b = (int)(k & 1);
            while (flags[c * 2 + b] != 0)
                __builtin_amdgcn_s_sleep(1);
            asm volatile("" ::: "memory");
            const size_t tt = ((size_t)blockIdx.x * CONS + c) * q + k;
            const char *gc = reinterpret_cast<const char *>(col + tt * T) + lane * 16;
            const char *gv = reinterpret_cast<const char *>(val + tt * T) + lane * 16;
            char *st = stages + (c * 2 + b) * STAGE;
#pragma unroll
            for (int i = 0; i < 2; i++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gc + i * 1024),
                                                 (__attribute__((address_space(3))) void *)(st + i * 1024), 16, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; i++)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gv + i * 1024),
                                                 (__attribute__((address_space(3))) void *)(st + 2048 + i * 1024), 16, 0, 0);
            if (n >= (size_t)(D - 1)) { // the transfer issued D - 1 steps ago has landed: hand its buffer over
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * 6) : "memory");
                int c2; size_t k2;
                target(n - (D - 1), c2, k2);
                flags[c2 * 2 + (int)(k2 & 1)] = 1;
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (size_t n = total > (size_t)(D - 1) ? total - (D - 1) : 0; n < total; n++) {
            int c2; size_t k2;
            target(n, c2, k2);
            flags[c2 * 2 + (int)(k2 & 1)] = 1;
        }
        return;
    }
    const auto xbuf = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x) + (size_t)xcd * (xbytes_per_xcd / 8), (short)0,
                                                         xbytes_per_xcd, 0x00020000);
    double acc = 0;
    for (size_t k = 0; k < q; k++) {
        const int b = (int)(k & 1);
        while (flags[wave * 2 + b] == 0)
            __builtin_amdgcn_s_sleep(1);
        asm volatile("" ::: "memory");
        const char *st = stages + (wave * 2 + b) * STAGE;
        const int *sc = reinterpret_cast<const int *>(st);
        const double *sv = reinterpret_cast<const double *>(st + 2048);
        int c[SIGMA];
        double v[SIGMA];
#pragma unroll
        for (int i = 0; i < SIGMA; i++)
            c[i] = sc[i * 64 + lane];
#pragma unroll
        for (int i = 0; i < SIGMA; i++)
            v[i] = sv[i * 64 + lane];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        flags[wave * 2 + b] = 0;
        unsigned long long g[SIGMA];
#pragma unroll
        for (int i = 0; i < SIGMA; i++) {
            const unsigned off = c[i] < 0 ? 0xFFFFFFFFu : (unsigned)c[i] * 8u;
            g[i] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 0));
        }
        double s = 0;
#pragma unroll
        for (int i = 0; i < SIGMA; i++) {
            const unsigned long long tw = __builtin_bit_cast(unsigned long long, hot[c[i] < 0 ? (unsigned)c[i] & 0x7FFFFFFFu : 0u]);
            s = __builtin_fma(v[i], __builtin_bit_cast(double, g[i] | tw), s);
        }
#pragma unroll
        for (int k2 = 0; k2 < WORK; k2++) {
            const unsigned long long bb = __builtin_bit_cast(unsigned long long, s);
            const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)bb, 0x111, 0xF, 0xF, true);
            const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(bb >> 32), 0x111, 0xF, 0xF, true);
            s += __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo) * 1e-30;
        }
        acc += s;
    }
    if (acc == 0.12345)
        out[((size_t)blockIdx.x * CONS + wave) * 64 + lane] = acc;
}

template <int PROD>
static int run_split(const char *name, const int *col, const double *val, const double *x, int xb, size_t ntiles, double *out, hipStream_t s)
{
    auto kern = k_split<PROD>;
    const int lds = TABLE * 8 + CONS * 2 * STAGE + 256;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(a, s));
        hipLaunchKernelGGL(kern, dim3(256), dim3((CONS + PROD) * 64), lds, s, col, val, x, xb, ntiles, out);
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    printf("%-52s %8.1f us\n", name, best * 1e3);
    fflush(stdout);
    return 0;
}

template <int MODE>
static int run(const char *name, const int *col, const double *val, const double *x, int xb, size_t ntiles, double *out, hipStream_t s, int coldpct)
{
    auto kern = k_probe<MODE>;
    const int lds = TABLE * 8 + WAVES * STAGE;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(a, s));
        hipLaunchKernelGGL(kern, dim3(256), dim3(WAVES * 64), lds, s, col, val, x, xb, ntiles, out, coldpct, (unsigned)(xb / 8));
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    printf("%-52s %8.1f us\n", name, best * 1e3);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv)
{
    const size_t nnz = argc > 1 ? (size_t)atoll(argv[1]) : ((size_t)1 << 28);
    const char *colds = argc > 2 ? argv[2] : "0,14,28";
    const int xkb = argc > 3 ? atoi(argv[3]) : 3600;
    const size_t ntiles = nnz / T;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    int *col;
    double *val, *x, *out;
    const int xb = xkb * 1024;
#ifdef INTERLEAVE
    CK(hipMalloc(&col, nnz * 12));
    val = nullptr;
#else
    CK(hipMalloc(&col, nnz * 4));
    CK(hipMalloc(&val, nnz * 8));
#endif
    CK(hipMalloc(&x, (size_t)xb * 8));
    CK(hipMemset(x, 0, (size_t)xb * 8));
    CK(hipMalloc(&out, (size_t)4096 * 64 * 8));
    for (const char *q = colds; *q;) {
        const int coldpct = atoi(q);
        while (*q && *q != ',') q++;
        if (*q == ',') q++;
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s, col, val, nnz, coldpct, xb / 8);
        CK(hipStreamSynchronize(s));
        printf("## %zu elements, %zu tiles of %d, %d %% of the gather lanes cold (x region %d KB per XCD), 8 wavefronts per CU, WORK %d\n", nnz, ntiles, T, coldpct, xkb, WORK);
        if (run<0>("streams through registers + gathers", col, val, x, xb, ntiles, out, s, coldpct)) return 1;
#ifndef INTERLEAVE
        if (run<1>("streams through LDS-DMA + gathers", col, val, x, xb, ntiles, out, s, coldpct)) return 1;
#endif
        if (run<2>("no streams (codes computed) + gathers", col, val, x, xb, ntiles, out, s, coldpct)) return 1;
        if (run<3>("streams only (loaded and consumed, no gather instructions)", col, val, x, xb, ntiles, out, s, coldpct)) return 1;
#ifndef INTERLEAVE
        if (run_split<2>("8 consumers + 2 producers (LDS-DMA, 2 buffers each)", col, val, x, xb, ntiles, out, s)) return 1;
        if (run_split<4>("8 consumers + 4 producers", col, val, x, xb, ntiles, out, s)) return 1;
        if (run_split<8>("8 consumers + 8 producers", col, val, x, xb, ntiles, out, s)) return 1;
#endif
    }
    return 0;
}

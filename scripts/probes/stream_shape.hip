// Micro-probe (round 3): what does the SHAPE of the persistent slab kernel cost when nothing but its memory traffic is
// left?  256 workgroups x 1024 threads (one per CU, 160 KB of LDS claimed like k_spmv_hot), every wavefront walks
// 512-element tiles of a 268 M-element column (4 B) + value (8 B) stream -- the R-MAT 24 child -- and per tile does
//   8 column loads, 8 value loads (coalesced, non-temporal), 8 "x gathers" and one FMA per element.
// Knobs:
//   DEPTH   1 = load tile, wait, gather, wait, compute (k_spmv_hot today)
//           2 = the next tile's streams are issued right behind this tile's gathers
//           3 = ... and the next tile's gathers before this tile's compute
//   CONTIG  0 = tiles dealt round robin over the 4096 wavefronts, 1 = one contiguous range per wavefront
//   COLD    share (percent) of gather lanes that go to memory (uniform over an 8-MB x); the others read the LDS table
//   PSTORE  every tile also stores 76 partial sums (contiguous per tile) like the child's y flush
//   WORK    extra dependent VALU work per tile (stands in for decode / flag walk / segmented scan)
// Prints achieved GB/s of the 12 B/element streams.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

constexpr int SIGMA = 8, T = 64 * SIGMA, TABLE = 12288;

__global__ void k_fill(int *col, double *val, size_t n, int coldpct, int xcols)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned long long h = i * 0x9E3779B97F4A7C15ull;
        h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
        const bool cold = (int)(h % 100) < coldpct;
        const unsigned r = (unsigned)(h >> 20);
        col[i] = cold ? (int)(r % (unsigned)xcols) : (int)(0x80000000u | (1u + r % (TABLE - 1)));
        val[i] = 1.0;
    }
}

struct Tile {
    int c[SIGMA];
    double v[SIGMA];
};
struct Gath {
    unsigned long long g[SIGMA];
};

__device__ __forceinline__ void load_tile(Tile &r, const int *col, const double *val, size_t t, int lane)
{
    const int *ct = col + t * T + lane;
    const double *vt = val + t * T + lane;
#pragma unroll
    for (int i = 0; i < SIGMA; i++)
        r.c[i] = __builtin_nontemporal_load(ct + i * 64);
#pragma unroll
    for (int i = 0; i < SIGMA; i++)
        r.v[i] = __builtin_nontemporal_load(vt + i * 64);
}

template <int COLD>
__device__ __forceinline__ void gather(Gath &q, const Tile &r, __amdgpu_buffer_rsrc_t xbuf)
{
#pragma unroll
    for (int i = 0; i < SIGMA; i++) {
        if (COLD > 0) {
            const unsigned off = r.c[i] < 0 ? 0xFFFFFFFFu : (unsigned)r.c[i] * 8u;
            q.g[i] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 0));
        } else {
            q.g[i] = 0;
        }
    }
}

template <int WORK>
__device__ __forceinline__ double compute(const Tile &r, const Gath &q, const __attribute__((address_space(3))) double *hot)
{
    double s = 0;
#pragma unroll
    for (int i = 0; i < SIGMA; i++) {
        const unsigned long long tw = __builtin_bit_cast(unsigned long long, hot[r.c[i] < 0 ? (unsigned)r.c[i] & 0x7FFFFFFFu : 0u]);
        s = __builtin_fma(r.v[i], __builtin_bit_cast(double, q.g[i] | tw), s);
    }
    // stand-in for the per-tile lane / cross-lane work: WORK dependent DPP adds
#pragma unroll
    for (int k = 0; k < WORK; k++) {
        const unsigned long long b = __builtin_bit_cast(unsigned long long, s);
        const int lo = __builtin_amdgcn_update_dpp(0, (int)(unsigned)b, 0x111, 0xF, 0xF, true);
        const int hi = __builtin_amdgcn_update_dpp(0, (int)(unsigned)(b >> 32), 0x111, 0xF, 0xF, true);
        s += __builtin_bit_cast(double, ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo) * 1e-30;
    }
    return s;
}

template <int DEPTH, bool CONTIG, int COLD, int PSTORE, int WORK>
__global__ void __launch_bounds__(1024) k_stream(const int *__restrict__ col, const double *__restrict__ val,
                                                 const double *__restrict__ x, int xbytes, size_t ntiles,
                                                 double *__restrict__ P, double *__restrict__ out)
{
    extern __shared__ __attribute__((aligned(16))) char smem[];
    auto *hot = (__attribute__((address_space(3))) double *)(smem);
    for (int j = threadIdx.x; j < TABLE; j += 1024)
        hot[j] = j ? 1.0 : 0.0;
    __syncthreads();
    const auto xbuf = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x), (short)0, xbytes, 0x00020000);
    const int lane = threadIdx.x & 63;
    const size_t nw = (size_t)gridDim.x * 16, w = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6);
    size_t t, t1, step;
    if (CONTIG) {
        const size_t q = ntiles / nw, rem = ntiles % nw;
        t = w * q + (w < rem ? w : rem);
        t1 = t + q + (w < rem ? 1 : 0);
        step = 1;
    } else {
        t = w;
        t1 = ntiles;
        step = nw;
    }
    double acc = 0;
    // PSTORE: 0 none, 1 plain (76 values per tile at a 608-byte stride), 2 non-temporal, 3 line-aligned (128 values per
    // tile), 4 plain but only every 8th tile (8 tiles' worth at once), 5 plain into a 1-MB window per wavefront group
    // (stays in L2), 6 write-through (sc1)
    double held[8];
    int nheld = 0;
    auto finish = [&](size_t tt, double s) {
        acc += s;
        if (PSTORE == 1) {
            P[tt * 76 + lane] = s;
            if (lane < 12)
                P[tt * 76 + 64 + lane] = s;
        } else if (PSTORE == 2) {
            __builtin_nontemporal_store(s, P + tt * 76 + lane);
            if (lane < 12)
                __builtin_nontemporal_store(s, P + tt * 76 + 64 + lane);
        } else if (PSTORE == 3) {
            P[tt * 128 + lane] = s;
            P[tt * 128 + 64 + lane] = s;
        } else if (PSTORE == 4) {
            held[nheld & 7] = s;
            nheld++;
            if ((nheld & 7) == 0) {
#pragma unroll
                for (int h = 0; h < 8; h++) {
                    P[(tt - 7 + h) * 76 + lane] = held[h];
                    if (lane < 12)
                        P[(tt - 7 + h) * 76 + 64 + lane] = held[h];
                }
            }
        } else if (PSTORE == 5) {
            const size_t o = (tt * 76) & ((1u << 17) - 1);
            P[o + lane] = s;
            if (lane < 12)
                P[o + 64 + lane] = s;
        } else if (PSTORE == 6) {
            __hip_atomic_store(P + tt * 76 + lane, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (lane < 12)
                __hip_atomic_store(P + tt * 76 + 64 + lane, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    };
    if (DEPTH == 1) {
        for (; t < t1; t += step) {
            Tile a;
            Gath ga;
            load_tile(a, col, val, t, lane);
            __builtin_amdgcn_sched_barrier(0);
            gather<COLD>(ga, a, xbuf);
            __builtin_amdgcn_sched_barrier(0);
            finish(t, compute<WORK>(a, ga, hot));
        }
    } else if (DEPTH == 2) {
        Tile a, b;
        Gath ga;
        if (t < t1)
            load_tile(a, col, val, t, lane);
        for (; t < t1; t += 2 * step) {
            __builtin_amdgcn_sched_barrier(0);
            gather<COLD>(ga, a, xbuf);
            __builtin_amdgcn_sched_barrier(0);
            load_tile(b, col, val, t + step < t1 ? t + step : t, lane);
            __builtin_amdgcn_sched_barrier(0);
            finish(t, compute<WORK>(a, ga, hot));
            if (t + step >= t1)
                break;
            __builtin_amdgcn_sched_barrier(0);
            gather<COLD>(ga, b, xbuf);
            __builtin_amdgcn_sched_barrier(0);
            load_tile(a, col, val, t + 2 * step < t1 ? t + 2 * step : t, lane);
            __builtin_amdgcn_sched_barrier(0);
            finish(t + step, compute<WORK>(b, ga, hot));
        }
    } else {
        // DEPTH 3: the gathers of tile k+1 and the streams of tile k+2 are in flight while tile k computes.  Three
        // stream register sets and two gather sets rotate: six steps bring every name back to its role.
        Tile a, b, c;
        Gath ga, gb;
        auto clampt = [&](size_t tt) { return tt < t1 ? tt : (t1 - 1); };
        if (t < t1) {
            load_tile(a, col, val, t, lane);
            load_tile(b, col, val, clampt(t + step), lane);
            __builtin_amdgcn_sched_barrier(0);
            gather<COLD>(ga, a, xbuf);
        }
#define STEP3(X, G, Y, H, Z, J)                                                                    \
    {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        gather<COLD>(H, Y, xbuf);                                                                  \
        load_tile(Z, col, val, clampt(t + ((J) + 2) * step), lane);                                \
        __builtin_amdgcn_sched_barrier(0);                                                         \
        finish(t + (J) * step, compute<WORK>(X, G, hot));                                          \
        if (t + ((J) + 1) * step >= t1)                                                            \
            break;                                                                                 \
    }
        for (; t < t1; t += 6 * step) {
            STEP3(a, ga, b, gb, c, 0)
            STEP3(b, gb, c, ga, a, 1)
            STEP3(c, ga, a, gb, b, 2)
            STEP3(a, gb, b, ga, c, 3)
            STEP3(b, ga, c, gb, a, 4)
            STEP3(c, gb, a, ga, b, 5)
        }
#undef STEP3
    }
    if (acc == 0.12345)
        out[w * 64 + lane] = acc;
}

static int g_copies = 1;

template <int DEPTH, bool CONTIG, int COLD, int PSTORE, int WORK>
static int run(const char *name, const int *col0, const double *val0, const double *x, int xbytes, size_t ntiles, double *P,
               double *out, hipStream_t s)
{
    static int turn = 0;
    const int *col = col0;
    const double *val = val0;
    auto kern = k_stream<DEPTH, CONTIG, COLD, PSTORE, WORK>;
    const int lds = 160 * 1024;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(kern, dim3(256), dim3(1024), lds, s, col, val, x, xbytes, ntiles, P, out);
    CK(hipStreamSynchronize(s));
    float best = 1e30f;
    for (int r = 0; r < 4 * g_copies; r++) {
        turn = (turn + 1) % g_copies; // cold protocol: every launch streams another copy
        col = col0 + (size_t)turn * ntiles * T;
        val = val0 + (size_t)turn * ntiles * T;
        CK(hipEventRecord(a, s));
        hipLaunchKernelGGL(kern, dim3(256), dim3(1024), lds, s, col, val, x, xbytes, ntiles, P, out);
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    const double bytes = (double)ntiles * T * 12.0;
    printf("%-44s %8.1f us   %7.1f GB/s of col+val streams\n", name, best * 1e3, bytes / (best * 1e-3) / 1e9);
    fflush(stdout);
    return 0;
}

int main(int argc, char **argv)
{
    const size_t nnz = argc > 1 ? (size_t)atoll(argv[1]) : ((size_t)1 << 28);
    g_copies = argc > 2 ? atoi(argv[2]) : 1; // > 1: rotate over that many copies of the streams (cold-cache protocol)
    const bool brief = argc > 3;
    const size_t ntiles = nnz / T;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    int *col;
    double *val, *x, *P, *out;
    const int xbytes = 8 << 20;
    CK(hipMalloc(&col, nnz * 4 * g_copies));
    CK(hipMalloc(&val, nnz * 8 * g_copies));
    CK(hipMalloc(&x, xbytes));
    CK(hipMemset(x, 0, xbytes));
    CK(hipMalloc(&P, ntiles * 128 * 8));
    CK(hipMalloc(&out, (size_t)4096 * 64 * 8));
    for (int coldpct : {0, 34}) {
        hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, s, col, val, nnz * g_copies, coldpct, xbytes / 8);
        CK(hipStreamSynchronize(s));
        printf("## %zu elements, %zu tiles of %d, cold gather lanes %d %% (x = 8 MB uniform)\n", nnz, ntiles, T, coldpct);
#define RUN(D, C, CO, PS, W) \
    if (run<D, C, CO, PS, W>("depth " #D " contig " #C " cold " #CO " pstore-mode " #PS " work " #W, col, val, x, xbytes, ntiles, P, out, s)) return 1;
        if (brief) { // small-matrix question: does one-tile-ahead prefetch help a 20-tile-per-wavefront kernel, cold?
            if (coldpct == 0) {
                RUN(1, true, 0, 0, 0) RUN(2, true, 0, 0, 0) RUN(1, true, 0, 1, 300) RUN(2, true, 0, 1, 300) RUN(3, true, 0, 1, 300)
            }
            continue;
        }
        if (coldpct == 0) {
            RUN(1, true, 0, 0, 0) RUN(1, true, 0, 1, 0) RUN(1, true, 0, 2, 0) RUN(1, true, 0, 3, 0) RUN(1, true, 0, 4, 0)
            RUN(1, true, 0, 5, 0) RUN(1, true, 0, 6, 0)
            RUN(2, true, 0, 1, 0) RUN(3, true, 0, 1, 0) RUN(3, true, 0, 2, 0) RUN(3, true, 0, 4, 0)
            RUN(1, true, 0, 0, 300) RUN(1, true, 0, 1, 300) RUN(1, true, 0, 2, 300) RUN(1, true, 0, 4, 300) RUN(1, true, 0, 5, 300)
            RUN(2, true, 0, 0, 300) RUN(2, true, 0, 1, 300) RUN(2, true, 0, 4, 300)
        } else {
            RUN(1, true, 34, 0, 300) RUN(1, true, 34, 1, 300) RUN(1, true, 34, 4, 300) RUN(2, true, 34, 0, 300) RUN(2, true, 34, 1, 300)
        }
    }
    return 0;
}

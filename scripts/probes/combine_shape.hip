// Micro-probe (round 3): what bounds the slab combine?  Synthetic partial sums of the R-MAT 24 shape: m rows, S slabs,
// every (row, slab) pair present with probability p (hub-free), P stored slab after slab and row after row inside a slab.
// Variants of "one wavefront sums the partials of ROWS consecutive rows":
//   RUNS   : (product, round 3) per (block, slab) one load of the run's partials + one load of its row bytes, LDS adds
//   NOIDX  : RUNS without the row-byte loads (rows taken from the lane id: wrong sums, same traffic otherwise)
//   NOLDS  : RUNS without the LDS read-modify-write (partials summed in a register)
//   DENSE  : the block's partials as ONE contiguous region (what a block-major P would offer): dense loads + LDS adds
//   STREAM : read P and write y, nothing else (the floor)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum { RUNS = 0, NOIDX = 1, NOLDS = 2, DENSE = 3, STREAM = 4 };

template <int MODE, int S, int ROWS, typename IDX>
__global__ void __launch_bounds__(256)
k_combine(int m, int m2, const unsigned *__restrict__ base, const IDX *__restrict__ rowidx,
          const unsigned *__restrict__ nonempty, const double *__restrict__ P, double *__restrict__ y,
          const unsigned *__restrict__ B0, const IDX *__restrict__ idx2, const double *__restrict__ P2)
{
    __shared__ double acc_all[4][ROWS + 64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int blk = blockIdx.x * 4 + w;
    const int r0 = blk * ROWS;
    if (r0 >= m)
        return;
    double *acc = acc_all[w];
    unsigned bw = base[(size_t)blk * S + (lane < 2 * S ? lane : 0)];
    unsigned ne[ROWS / 64];
#pragma unroll
    for (int j = 0; j < ROWS / 64; j++)
        ne[j] = nonempty[(r0 >> 5) + 2 * j + (lane >> 5)];
#pragma unroll
    for (int j = 0; j < ROWS / 64; j++)
        acc[j * 64 + lane] = 0;
    const unsigned dummy = ROWS + lane;
    double reg = 0;
    if (MODE == STREAM || MODE == DENSE) {
        // block-major layout: the block's partials are ONE contiguous region [B0[blk], B0[blk + 1]) ordered by (slab, row);
        // dense loads, then -- chunk by chunk -- one masked LDS pass per run that intersects the chunk (inside a run every
        // row occurs once; two runs of one chunk may hold the same row)
        const int lo = (int)B0[blk], total = (int)B0[blk + 1] - lo;
        int rb[S + 1]; // run boundaries relative to the region
        rb[0] = 0;
#pragma unroll
        for (int k = 0; k < S; k++)
            rb[k + 1] = rb[k] + __builtin_amdgcn_readlane((int)bw, S + k) - __builtin_amdgcn_readlane((int)bw, k);
        for (int off = 0; off < total; off += 64 * 4) {
            double part[4];
            unsigned idx[4];
#pragma unroll
            for (int q = 0; q < 4; q++) {
                int j = off + q * 64 + lane;
                j = lo + (j < total ? j : total - 1);
                j = j < 0 ? 0 : j;
                part[q] = P2[j];
                idx[q] = MODE == DENSE ? (unsigned)idx2[j] : 0u;
            }
#pragma unroll
            for (int q = 0; q < 4; q++) {
                if (MODE == DENSE) {
                    const int c0 = off + q * 64, pos = c0 + lane;
#pragma unroll
                    for (int k = 0; k < S; k++)
                        if (rb[k + 1] > c0 && rb[k] < c0 + 64) { // (wave-uniform)
                            const unsigned slot = pos >= rb[k] && pos < rb[k + 1] ? (idx[q] & (ROWS - 1)) : dummy;
                            acc[slot] += part[q];
                        }
                } else {
                    reg += part[q];
                }
            }
        }
    } else {
        int lo[S], len[S], longest = 0;
#pragma unroll
        for (int q = 0; q < S; q++) {
            lo[q] = __builtin_amdgcn_readlane((int)bw, q);
            len[q] = __builtin_amdgcn_readlane((int)bw, S + q) - lo[q];
            longest = len[q] > longest ? len[q] : longest;
        }
        for (int off = 0; off < longest; off += 64) {
            double part[S];
            unsigned idx[S];
#pragma unroll
            for (int q = 0; q < S; q++) {
                int j = off + lane < len[q] ? off + lane : len[q] - 1;
                j += lo[q];
                j = j < 0 ? 0 : (j < m2 ? j : m2 - 1);
                part[q] = P[j];
                idx[q] = MODE == NOIDX ? (unsigned)lane : (unsigned)rowidx[j];
            }
#pragma unroll
            for (int q = 0; q < S; q++) {
                if (MODE == NOLDS) {
                    reg += off + lane < len[q] ? part[q] * (double)idx[q] : 0.0;
                } else {
                    const unsigned slot = off + lane < len[q] ? (idx[q] & (ROWS - 1)) : dummy;
                    acc[slot] += part[q];
                }
            }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int j = 0; j < ROWS / 64; j++) {
        const int r = r0 + j * 64 + lane;
        if (r < m && ((ne[j] >> (lane & 31)) & 1u))
            y[r] = acc[j * 64 + lane] + reg;
    }
}

// ---- synthetic structure -------------------------------------------------------------------------------------------------
__device__ __forceinline__ bool present(unsigned r, unsigned k, unsigned thr)
{
    unsigned long long h = ((unsigned long long)r << 8 | k) * 0x9E3779B97F4A7C15ull;
    h ^= h >> 31; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 29;
    return (unsigned)(h >> 40) % 1000u < thr;
}
// count[b * S + k] = segments of slab k in row block b
template <int ROWS>
__global__ void k_count(int m, int S, unsigned thr, unsigned *count)
{
    const int b = blockIdx.x, k = threadIdx.x;
    if (k >= S)
        return;
    unsigned c = 0;
    for (int r = b * ROWS; r < (b + 1) * ROWS && r < m; r++)
        c += present(r, k, thr);
    count[(size_t)b * S + k] = c;
}
// base[b][k] from the per-slab prefix of count (host does the S column scans; tiny)
template <int ROWS, typename IDX>
__global__ void k_fill(int m, int S, unsigned thr, const unsigned *base, IDX *rowidx, double *P, unsigned *nonempty)
{
    const int b = blockIdx.x, k = threadIdx.x;
    if (k < S) {
        unsigned s = base[(size_t)b * S + k];
        for (int r = b * ROWS; r < (b + 1) * ROWS && r < m; r++)
            if (present(r, k, thr)) {
                rowidx[s] = (IDX)(r & (ROWS - 1));
                P[s] = 1.0;
                s++;
            }
    }
    if (k == 0)
        for (int r = b * ROWS; r < (b + 1) * ROWS && r < m; r += 32) {
            unsigned bits = 0;
            for (int i = 0; i < 32 && r + i < m; i++) {
                bool any = false;
                for (int kk = 0; kk < S; kk++)
                    any |= present(r + i, kk, thr);
                bits |= (unsigned)any << i;
            }
            nonempty[r >> 5] = bits;
        }
}

// block-major copies: region of block b = [B0[b], B0[b + 1]), ordered by (slab, row)
template <int ROWS, typename IDX>
__global__ void k_fill2(int m, int S, unsigned thr, const unsigned *B0, IDX *idx2, double *P2)
{
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if ((long long)b * ROWS >= m)
        return;
    unsigned s = B0[b];
    for (int k = 0; k < S; k++)
        for (int r = b * ROWS; r < (b + 1) * ROWS && r < m; r++)
            if (present(r, k, thr)) {
                idx2[s] = (IDX)(r & (ROWS - 1));
                P2[s] = 1.0;
                s++;
            }
}

template <int S, int ROWS, typename IDX>
static int shape(int m, unsigned thr, hipStream_t s)
{
    const int nblk = (m + ROWS - 1) / ROWS;
    unsigned *count, *base, *nonempty;
    CK(hipMalloc(&count, (size_t)(nblk + 2) * S * 4));
    CK(hipMalloc(&base, (size_t)(nblk + 2) * S * 4));
    CK(hipMalloc(&nonempty, ((size_t)m / 32 + 64) * 4));
    hipLaunchKernelGGL(k_count<ROWS>, dim3(nblk), dim3(64), 0, s, m, S, thr, count);
    std::vector<unsigned> hc((size_t)(nblk + 2) * S, 0u), hb((size_t)(nblk + 2) * S, 0u);
    CK(hipStreamSynchronize(s));
    CK(hipMemcpy(hc.data(), count, (size_t)nblk * S * 4, hipMemcpyDeviceToHost));
    unsigned run = 0;
    for (int k = 0; k < S; k++) {
        for (int b = 0; b < nblk; b++) {
            hb[(size_t)b * S + k] = run;
            run += hc[(size_t)b * S + k];
        }
        hb[(size_t)nblk * S + k] = run;
    }
    const int m2 = (int)run;
    CK(hipMemcpy(base, hb.data(), hb.size() * 4, hipMemcpyHostToDevice));
    std::vector<unsigned> hB0((size_t)nblk + 2, 0u);
    for (int b = 0; b < nblk; b++) {
        unsigned c = 0;
        for (int k = 0; k < S; k++)
            c += hc[(size_t)b * S + k];
        hB0[b + 1] = hB0[b] + c;
    }
    unsigned *B0;
    CK(hipMalloc(&B0, hB0.size() * 4));
    CK(hipMemcpy(B0, hB0.data(), hB0.size() * 4, hipMemcpyHostToDevice));
    IDX *rowidx, *idx2;
    double *P, *y, *P2;
    CK(hipMalloc(&idx2, ((size_t)m2 + 64) * sizeof(IDX)));
    CK(hipMalloc(&P2, ((size_t)m2 + 64) * 8));
    CK(hipMalloc(&rowidx, ((size_t)m2 + 64) * sizeof(IDX)));
    CK(hipMalloc(&P, ((size_t)m2 + 64) * 8));
    CK(hipMalloc(&y, (size_t)m * 8));
    hipLaunchKernelGGL((k_fill<ROWS, IDX>), dim3(nblk), dim3(64), 0, s, m, S, thr, base, rowidx, P, nonempty);
    hipLaunchKernelGGL((k_fill2<ROWS, IDX>), dim3((nblk + 63) / 64), dim3(64), 0, s, m, S, thr, B0, idx2, P2);
    CK(hipStreamSynchronize(s));
    printf("## m = %d, S = %d, rows per wavefront = %d, segments = %d (%.2f per row), row index = %zu B\n", m, S, ROWS, m2,
           (double)m2 / m, sizeof(IDX));
    const char *names[] = {"RUNS", "NOIDX", "NOLDS", "DENSE", "STREAM"};
    for (int mode = 0; mode < 5; mode++) {
        auto launch = [&]() {
            const dim3 grid((nblk + 3) / 4), block(256);
            switch (mode) {
            case 0: hipLaunchKernelGGL((k_combine<RUNS, S, ROWS, IDX>), grid, block, 0, s, m, m2, base, rowidx, nonempty, P, y, B0, idx2, P2); break;
            case 1: hipLaunchKernelGGL((k_combine<NOIDX, S, ROWS, IDX>), grid, block, 0, s, m, m2, base, rowidx, nonempty, P, y, B0, idx2, P2); break;
            case 2: hipLaunchKernelGGL((k_combine<NOLDS, S, ROWS, IDX>), grid, block, 0, s, m, m2, base, rowidx, nonempty, P, y, B0, idx2, P2); break;
            case 3: hipLaunchKernelGGL((k_combine<DENSE, S, ROWS, IDX>), grid, block, 0, s, m, m2, base, rowidx, nonempty, P, y, B0, idx2, P2); break;
            default: hipLaunchKernelGGL((k_combine<STREAM, S, ROWS, IDX>), grid, block, 0, s, m, m2, base, rowidx, nonempty, P, y, B0, idx2, P2); break;
            }
        };
        launch();
        CK(hipStreamSynchronize(s));
        hipEvent_t a, b;
        CK(hipEventCreate(&a));
        CK(hipEventCreate(&b));
        float best = 1e30f;
        for (int r = 0; r < 5; r++) {
            CK(hipEventRecord(a, s));
            launch();
            CK(hipEventRecord(b, s));
            CK(hipEventSynchronize(b));
            float ms;
            CK(hipEventElapsedTime(&ms, a, b));
            best = ms < best ? ms : best;
        }
        printf("   %-7s %8.1f us   (%.0f GB/s of 8 B per segment + 8 B per row)\n", names[mode], best * 1e3,
               ((double)m2 * 8 + (double)m * 8) / (best * 1e-3) / 1e9);
        fflush(stdout);
    }
    CK(hipFree(count)); CK(hipFree(base)); CK(hipFree(nonempty)); CK(hipFree(rowidx)); CK(hipFree(P)); CK(hipFree(y)); CK(hipFree(B0)); CK(hipFree(idx2)); CK(hipFree(P2));
    return 0;
}

int main()
{
    hipStream_t s;
    CK(hipStreamCreate(&s));
    const int m = 1 << 24;
    if (shape<16, 256, unsigned char>(m, 150, s)) return 1;   // 2.4 segments per row (R-MAT 24, 16 slabs)
    if (shape<16, 1024, unsigned short>(m, 150, s)) return 1;
    if (shape<32, 256, unsigned char>(m, 105, s)) return 1;   // 3.35 per row (32 slabs)
    if (shape<32, 1024, unsigned short>(m, 105, s)) return 1;
    return 0;
}

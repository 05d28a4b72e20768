// Micro-probe: rate of fp64 atomic adds (global_atomic_add_f64, no return) onto a y of 16.7 M doubles from the
// persistent shape of k_spmv_hot (256 workgroups x 1024 threads).  Pattern of a segment flush: one wave instruction
// adds 64 partial sums to rows r0 + lane * stride (+ optional random jitter), r0 random per instruction.
// Several XCDs hit the same rows (one slab per XCD): `--same` makes all waves with the same index modulo 512 share r0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void __launch_bounds__(1024) k(double *__restrict__ y, const unsigned *__restrict__ starts, int iters, int stride, int m,
                                          int plain_store)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6);
    const unsigned *st = starts + wave * iters;
    for (int it = 0; it < iters; it++) {
        size_t r = (size_t)st[it] + (size_t)lane * stride;
        r = r < (size_t)m ? r : r - m;
        if (plain_store)
            y[r] = 1.0;
        else
            unsafeAtomicAdd(&y[r], 1.0);
    }
}

int main()
{
    const int m = 1 << 24, waves = 256 * 16, iters = 224; // 58.7 M atomics
    double *y; CK(hipMalloc(&y, (size_t)m * 8)); CK(hipMemset(y, 0, (size_t)m * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    for (int shared = 0; shared < 2; shared++)
        for (int stride : {1, 2, 4, 8, 64, 4099}) {
            std::vector<unsigned> hs((size_t)waves * iters);
            unsigned long long stt = 88172645463325252ull;
            auto rnd = [&]() { stt ^= stt << 13; stt ^= stt >> 7; stt ^= stt << 17; return stt; };
            for (size_t w = 0; w < (size_t)waves; w++)
                for (int it = 0; it < iters; it++) {
                    // flush pattern: a wave walks rows upward; `shared`: the 8 XCDs (wave index modulo 512 equal) hit the same rows
                    const size_t owner = shared ? w % 512 : w;
                    unsigned long long h = owner * 1000003ull + (unsigned long long)it * 7919ull;
                    h ^= h >> 13; h *= 0x9E3779B97F4A7C15ull; h ^= h >> 29;
                    hs[w * iters + it] = (unsigned)(h % (unsigned)m);
                }
            (void)rnd;
            unsigned *starts; CK(hipMalloc(&starts, hs.size() * 4));
            CK(hipMemcpy(starts, hs.data(), hs.size() * 4, hipMemcpyHostToDevice));
            float ms[2];
            for (int plain = 0; plain < 2; plain++) {
                hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, s, y, starts, iters, stride, m, plain);
                CK(hipStreamSynchronize(s));
                hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
                CK(hipEventRecord(a, s));
                for (int r = 0; r < 3; r++) hipLaunchKernelGGL(k, dim3(256), dim3(1024), 0, s, y, starts, iters, stride, m, plain);
                CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
                CK(hipEventElapsedTime(&ms[plain], a, b)); ms[plain] /= 3;
            }
            const double n = (double)waves * iters * 64;
            printf("%s rows, lane stride %5d: atomic add %8.1f us (%.3f ns each, %.1f G/s)   plain store %8.1f us\n",
                   shared ? "SHARED by 8 waves" : "private         ", stride, ms[0] * 1e3, ms[0] * 1e6 / n, n / ms[0] / 1e6, ms[1] * 1e3);
            CK(hipFree(starts));
        }
    return 0;
}

// Micro-probe (round 3): can cold x gathers go through the SCALAR data cache (s_load, 64-B lines, its own fill path)
// instead of / next to the vector L1 (128-B line fill per cold lane = 0.88 ns per lane per CU on gfx950)?
// 256 workgroups x 1024 threads; every wavefront does ITER rounds of 64 random 8-B reads inside an 8-MB window of x
// (one slab's columns).  KS of the 64 reads of a round are issued as scalar loads (readlane -> s_load_dwordx2), the
// other 64 - KS as one masked vector gather.  Prints ns per read per CU for every KS.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned mix(unsigned a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

template <int KS>
__global__ void __launch_bounds__(1024)
k_gather(const double *__restrict__ x, unsigned xmask, int iters, double *__restrict__ out)
{
    const int lane = threadIdx.x & 63;
    const unsigned wid = blockIdx.x * 16 + (threadIdx.x >> 6);
    const double *xw = x + (size_t)(blockIdx.x & 7) * ((size_t)xmask + 1); // one window per XCD
    double acc = 0.0, sacc = 0.0;
    for (int it = 0; it < iters; it++) {
        const unsigned idx = mix(wid * 0x9E3779B9u + it * 64u + lane) & xmask;
        double sv[KS > 0 ? KS : 1];
#pragma unroll
        for (int l = 0; l < KS; l++) {
            const unsigned si = __builtin_amdgcn_readlane(idx, l);
            sv[l] = xw[si];
        }
        if (lane >= KS)
            acc += xw[idx];
#pragma unroll
        for (int l = 0; l < KS; l++)
            sacc += sv[l];
    }
    if (lane == 0)
        acc += sacc;
    out[(size_t)blockIdx.x * 1024 + threadIdx.x] = acc;
}

template <int KS>
static int run(const double *x, unsigned xmask, int iters, double *out)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    hipLaunchKernelGGL(k_gather<KS>, dim3(256), dim3(1024), 0, 0, x, xmask, iters, out);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 5; r++) {
        CK(hipEventRecord(a));
        hipLaunchKernelGGL(k_gather<KS>, dim3(256), dim3(1024), 0, 0, x, xmask, iters, out);
        CK(hipEventRecord(b));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    const double reads_per_cu = 16.0 * iters * 64.0;
    printf("KS=%2d  %8.1f us   %.3f ns per read per CU   (scalar %.3f ns each if the vector part were free)\n", KS, best * 1e3,
           best * 1e6 / reads_per_cu, KS ? best * 1e6 / (16.0 * iters * KS) : 0.0);
    return 0;
}

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    const unsigned xmask = (1u << 20) - 1; // 1 M doubles = 8 MB per window
    double *x, *out;
    CK(hipMalloc(&x, 8 * ((size_t)xmask + 1) * 8));
    CK(hipMemset(x, 0, 8 * ((size_t)xmask + 1) * 8));
    CK(hipMalloc(&out, 256 * 1024 * 8));
    run<0>(x, xmask, iters, out);
    run<1>(x, xmask, iters, out);
    run<2>(x, xmask, iters, out);
    run<4>(x, xmask, iters, out);
    run<8>(x, xmask, iters, out);
    run<16>(x, xmask, iters, out);
    run<32>(x, xmask, iters, out);
    run<64>(x, xmask, iters, out);
    return 0;
}

// Micro-probe (round 4): how fast does the chip start and retire wavefronts that do (almost) nothing?  k_slab_combine on R-MAT 24 is
// 16 384 workgroups x 256 threads with 10 KB of LDS each, and with every load removed it still needs 121 us.
//   LDS_BYTES of static LDS per workgroup (touched once), one 4-byte load and one 8-byte store per thread optional.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int LDS_BYTES, int MODE>
__global__ void __launch_bounds__(256) k_shape(const unsigned *__restrict__ in, double *__restrict__ out, int n)
{
    __shared__ double lds[LDS_BYTES / 8 > 0 ? LDS_BYTES / 8 : 1];
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    double v = 0;
    if (LDS_BYTES > 0) {
        lds[threadIdx.x] = (double)threadIdx.x;
        __builtin_amdgcn_wave_barrier();
        v = lds[threadIdx.x ^ 1];
    }
    if (MODE >= 1) // one dependent load (the combine's run bounds)
        v += (double)in[i % (size_t)n];
    if (MODE >= 2) // one store per thread (the combine's y)
        out[i] = v;
    else if (v == 0.12345)
        out[i] = v;
}

template <int LDS_BYTES, int MODE>
static int run(const char *name, int wgs, const unsigned *in, double *out, int n, hipStream_t s)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float best = 1e30f;
    for (int r = 0; r < 6; r++) {
        CK(hipEventRecord(a, s));
        hipLaunchKernelGGL((k_shape<LDS_BYTES, MODE>), dim3(wgs), dim3(256), 0, s, in, out, n);
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        float ms;
        CK(hipEventElapsedTime(&ms, a, b));
        best = ms < best ? ms : best;
    }
    printf("%-64s %8.1f us\n", name, best * 1e3);
    return 0;
}

int main()
{
    const int wgs = 16384, n = 1 << 20;
    hipStream_t s;
    CK(hipStreamCreate(&s));
    unsigned *in;
    double *out;
    CK(hipMalloc(&in, (size_t)n * 4));
    CK(hipMemset(in, 0, (size_t)n * 4));
    CK(hipMalloc(&out, (size_t)wgs * 256 * 8));
    printf("## %d workgroups x 256 threads (65 536 wavefronts)\n", wgs);
    if (run<0, 0>("no LDS, nothing", wgs, in, out, n, s)) return 1;
    if (run<10240, 0>("10 KB LDS, nothing", wgs, in, out, n, s)) return 1;
    if (run<10240, 1>("10 KB LDS, one load", wgs, in, out, n, s)) return 1;
    if (run<10240, 2>("10 KB LDS, one load, one 8-byte store per thread (134 MB)", wgs, in, out, n, s)) return 1;
    if (run<0, 2>("no LDS, one load, one store", wgs, in, out, n, s)) return 1;
    if (run<40960, 2>("40 KB LDS, one load, one store", wgs, in, out, n, s)) return 1;
    return 0;
}

// Micro-probe: what limits the dependent x[col] gather of a scircuit-sized SpMV (958936 gathers)?
// Every variant: 1000 launches in one hipGraph, us per launch; subtract "stream" to get the gather cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

enum Mode { NONE, GLOBAL, LDSWIN, HALF_LDS };

template <typename XT, int SIGMA, int MODE>
__global__ void __launch_bounds__(256) k(const int *__restrict__ col, const double *__restrict__ val,
                                         const XT *__restrict__ x, double *__restrict__ y, int ntiles, int n)
{
    __shared__ XT win[MODE >= LDSWIN ? 4096 : 1];
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    int w0 = 0;
    if (MODE >= LDSWIN) { // stage a 4096-element window of x (coalesced), block-wide
        w0 = (int)(((size_t)blockIdx.x * 256 * SIGMA) % (size_t)(n - 4096));
        for (int i = threadIdx.x; i < 4096; i += 256) win[i] = x[w0 + i];
        __syncthreads();
    }
    if (t >= ntiles) return;
    const size_t base = (size_t)t * 64 * SIGMA + lane;
    int c[SIGMA]; double v[SIGMA];
#pragma unroll
    for (int i = 0; i < SIGMA; i++) { c[i] = col[base + i * 64]; v[i] = val[base + i * 64]; }
    double s = 0;
#pragma unroll
    for (int i = 0; i < SIGMA; i++) {
        double xv;
        if (MODE == NONE) xv = (double)c[i];
        else if (MODE == GLOBAL) xv = (double)x[c[i]];
        else if (MODE == LDSWIN) xv = (double)win[c[i] & 4095];
        else { unsigned d = (unsigned)(c[i] - w0); xv = d < 4096u ? (double)win[d] : (double)x[c[i]]; }
        s += v[i] * xv;
    }
    y[(size_t)t * 64 + lane] = s;
}

template <typename F>
int time_graph(const std::string &name, F launch, hipStream_t s, int reps)
{
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < reps; i++) launch(s);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    printf("%-46s %8.3f us/launch\n", name.c_str(), ms * 1e3 / reps);
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    return 0;
}

int main()
{
    const int nnz = 958936, n = 170998;
    int *col; double *val, *x, *y; float *xf;
    CK(hipMalloc(&col, (size_t)nnz * 4 + 65536)); CK(hipMalloc(&val, (size_t)nnz * 8 + 65536));
    CK(hipMalloc(&x, (size_t)n * 8)); CK(hipMalloc(&xf, (size_t)n * 4)); CK(hipMalloc(&y, (size_t)nnz * 8));
    CK(hipMemset(val, 0, (size_t)nnz * 8 + 65536)); CK(hipMemset(x, 0, (size_t)n * 8)); CK(hipMemset(xf, 0, (size_t)n * 4));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int reps = 1000;
    std::vector<int> hc(nnz + 16384);
    // column patterns: fraction `near` of entries within +-64 of the scaled diagonal, rest uniform
    for (int pat = 0; pat < 5; pat++) {
        const double near = pat == 0 ? 0.0 : pat == 1 ? 0.5 : pat == 2 ? 0.9 : pat == 3 ? 1.0 : -1.0;
        unsigned long long st = 88172645463325252ull;
        auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
        for (size_t i = 0; i < hc.size(); i++) {
            long long diag = (long long)((double)i / hc.size() * n);
            if (near < 0) hc[i] = (int)(rnd() % 2048);                       // tiny hot vector (L1-resident)
            else if ((rnd() % 1000) < near * 1000) { long long c = diag + (long long)(rnd() % 129) - 64; hc[i] = (int)(c < 0 ? 0 : c >= n ? n - 1 : c); }
            else hc[i] = (int)(rnd() % n);
        }
        CK(hipMemcpy(col, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
        std::string tag = near < 0 ? "hot2048" : "near=" + std::to_string(near).substr(0, 4);
#define RUN(XT, XP, S, M, NAME) { int nt = nnz / (64 * S); time_graph(tag + " " NAME " sigma=" #S, [&](hipStream_t st_) { \
        hipLaunchKernelGGL((k<XT, S, M>), dim3((nt + 3) / 4), dim3(256), 0, st_, col, val, XP, y, nt, n); }, s, reps); }
        if (pat == 0) { RUN(double, x, 8, NONE, "stream-only  ") }
        RUN(double, x, 8, GLOBAL, "global f64   ")
        RUN(float, xf, 8, GLOBAL, "global f32   ")
        RUN(double, x, 4, GLOBAL, "global f64   ")
        RUN(double, x, 16, GLOBAL, "global f64   ")
        if (pat == 0) { RUN(double, x, 8, LDSWIN, "lds-only f64 ") }
        RUN(double, x, 8, HALF_LDS, "window+global")
    }
    return 0;
}

// Micro-probe for the per-process step-time modes: in ONE process measure (a) the shader clock (cycles of
// s_memtime per 100-MHz wall_clock tick over a delay loop), (b) the cost of an empty dependent launch and
// (c) a scircuit-sized gather kernel, each replayed from a hipGraph.  Run it several times: if (b)/(c) move
// together with (a) the modes are clock states.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(int *out) { if (threadIdx.x == 9999) out[0] = 1; }
__global__ void k_clock(unsigned long long *out, int spin)
{
    const unsigned long long w0 = wall_clock64(), c0 = clock64();
    float v = threadIdx.x;
    for (int i = 0; i < spin; i++) v = v * 1.0001f + 0.5f;
    const unsigned long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
    if (v == -1.f) out[0] = 0;
}
template <int SIGMA>
__global__ void __launch_bounds__(256) k_gather(const int *__restrict__ col, const double *__restrict__ val,
                                                const double *__restrict__ x, double *__restrict__ y, int ntiles)
{
    const int lane = threadIdx.x & 63;
    const int t = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (t >= ntiles) return;
    const size_t base = (size_t)t * 64 * SIGMA + lane;
    int c[SIGMA]; double v[SIGMA];
#pragma unroll
    for (int i = 0; i < SIGMA; i++) { c[i] = col[base + i * 64]; v[i] = val[base + i * 64]; }
    double s = 0;
#pragma unroll
    for (int i = 0; i < SIGMA; i++) s += v[i] * x[c[i]];
    y[(size_t)t * 64 + lane] = s;
}
template <typename F>
double time_graph(F launch, hipStream_t s, int reps)
{
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
    for (int i = 0; i < reps; i++) launch(s);
    hipStreamEndCapture(s, &g);
    hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
    for (int w = 0; w < 3; w++) hipGraphLaunch(ge, s);
    hipStreamSynchronize(s);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a, s); hipGraphLaunch(ge, s); hipEventRecord(b, s); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms * 1e3 / reps;
}
int main()
{
    const int nnz = 958936, n = 170998;
    int *col; double *val, *x, *y; int *out; unsigned long long *clk;
    CK(hipMalloc(&col, (size_t)nnz * 4 + 65536)); CK(hipMalloc(&val, (size_t)nnz * 8 + 65536));
    CK(hipMalloc(&x, (size_t)n * 8)); CK(hipMalloc(&y, (size_t)nnz * 8)); CK(hipMalloc(&out, 64));
    CK(hipMalloc(&clk, 16 * 256));
    std::vector<int> hc(nnz + 16384); for (size_t i = 0; i < hc.size(); i++) hc[i] = (int)((i * 2654435761u) % n);
    CK(hipMemcpy(col, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(val, 0, (size_t)nnz * 8 + 65536)); CK(hipMemset(x, 0, (size_t)n * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int nt = nnz / (64 * 5);
    for (int round = 0; round < 3; round++) {
        const double e = time_graph([&](hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(750), dim3(256), 0, st, out); }, s, 1000);
        const double g = time_graph([&](hipStream_t st) { hipLaunchKernelGGL((k_gather<5>), dim3((nt + 3) / 4), dim3(256), 0, st, col, val, x, y, nt); }, s, 1000);
        hipLaunchKernelGGL(k_clock, dim3(256), dim3(64), 0, s, clk, 200000);
        CK(hipStreamSynchronize(s));
        unsigned long long h[512]; CK(hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost));
        double cyc = 0, wall = 0; for (int i = 0; i < 256; i++) { cyc += (double)h[2 * i]; wall += (double)h[2 * i + 1]; }
        printf("empty %.3f us  gather %.3f us  shader clock %.0f MHz\n", e, g, cyc / wall * 100.0);
    }
    return 0;
}

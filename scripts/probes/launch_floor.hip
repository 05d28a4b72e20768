// Micro-probe: what does one dependent launch cost on this chip for a scircuit-sized problem?
//   empty      : 750 x 256 threads, no memory traffic
//   stream     : reads 12 B/element for 958936 elements (coalesced), one 8-B store per 5 elements
//   gather     : + dependent gather x[col] from a 1.37 MB vector
// Each variant is captured 1000x into a hipGraph and replayed; prints us per launch.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(int *out) { if (threadIdx.x == 9999) out[0] = 1; }

template <int SIGMA, bool GATHER>
__global__ void __launch_bounds__(256) k_stream(const int *__restrict__ col, const double *__restrict__ val,
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
    for (int i = 0; i < SIGMA; i++) s += v[i] * (GATHER ? x[c[i]] : (double)c[i]);
    y[(size_t)t * 64 + lane] = s;
}

template <typename F>
int time_graph(const char *name, F launch, hipStream_t s, int reps)
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
    printf("%-28s %8.3f us/launch\n", name, ms * 1e3 / reps);
    return 0;
}

int main()
{
    const int nnz = 958936, n = 170998;
    int *col; double *val, *x, *y; int *out;
    CK(hipMalloc(&col, (size_t)nnz * 4 + 65536)); CK(hipMalloc(&val, (size_t)nnz * 8 + 65536));
    CK(hipMalloc(&x, (size_t)n * 8)); CK(hipMalloc(&y, (size_t)nnz * 8)); CK(hipMalloc(&out, 64));
    std::vector<int> hc(nnz + 16384); for (size_t i = 0; i < hc.size(); i++) hc[i] = (int)((i * 2654435761u) % n);
    CK(hipMemcpy(col, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(val, 0, (size_t)nnz * 8 + 65536)); CK(hipMemset(x, 0, (size_t)n * 8));
    hipStream_t s; CK(hipStreamCreate(&s));
    const int reps = 1000;
    for (int grid : {64, 256, 750, 3000})
        time_graph(("empty grid=" + std::to_string(grid)).c_str(), [&](hipStream_t st) { hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, st, out); }, s, reps);
#define RUN(S, G) { int nt = nnz / (64 * S); time_graph(G ? "gather sigma=" #S : "stream sigma=" #S, [&](hipStream_t st) { \
        hipLaunchKernelGGL((k_stream<S, G>), dim3((nt + 3) / 4), dim3(256), 0, st, col, val, x, y, nt); }, s, reps); }
    RUN(4, false) RUN(8, false) RUN(16, false) RUN(4, true) RUN(5, true) RUN(8, true) RUN(16, true) RUN(32, true)
    return 0;
}

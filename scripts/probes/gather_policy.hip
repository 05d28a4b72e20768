// Micro-probe: does the cache policy of the x gather change the cost of a random 8-byte gather?
// plain | nt (nontemporal) | sc1 (agent-scope relaxed atomic load) | sys (system-scope relaxed atomic load)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <string>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int SIGMA, int POLICY>
__global__ void __launch_bounds__(256) k(const int *__restrict__ col, const double *__restrict__ val,
                                         const double *x, double *__restrict__ y, int ntiles)
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
    for (int i = 0; i < SIGMA; i++) {
        double xv;
        if (POLICY == 0) xv = x[c[i]];
        else if (POLICY == 1) xv = __builtin_nontemporal_load(&x[c[i]]);
        else if (POLICY == 2) xv = __hip_atomic_load(&x[c[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else xv = __hip_atomic_load(&x[c[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
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

int main(int argc, char **argv)
{
    // sizes: scircuit-like (x = 1.4 MB) and a large one (x = 134 MB, 64 M gathers)
    for (int big = 0; big < 2; big++) {
        const size_t nnz = big ? (size_t)64 << 20 : 958936, n = big ? (size_t)16 << 20 : 170998;
        int *col; double *val, *x, *y;
        CK(hipMalloc(&col, nnz * 4 + 65536)); CK(hipMalloc(&val, nnz * 8 + 65536));
        CK(hipMalloc(&x, n * 8)); CK(hipMalloc(&y, nnz * 8 / 4));
        CK(hipMemset(val, 0, nnz * 8 + 65536)); CK(hipMemset(x, 0, n * 8));
        hipStream_t s; CK(hipStreamCreate(&s));
        const int reps = big ? 20 : 1000;
        std::vector<int> hc(nnz + 16384);
        unsigned long long st = 88172645463325252ull;
        for (size_t i = 0; i < hc.size(); i++) { st ^= st << 13; st ^= st >> 7; st ^= st << 17; hc[i] = (int)(st % n); }
        CK(hipMemcpy(col, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
        std::string tag = big ? "big(64M gathers, x=134MB) " : "small(0.96M gathers, x=1.4MB) ";
#define RUN(S, P, NAME) { int nt = (int)(nnz / (64 * S)); time_graph(tag + NAME " sigma=" #S, [&](hipStream_t st_) { \
        hipLaunchKernelGGL((k<S, P>), dim3((nt + 3) / 4), dim3(256), 0, st_, col, val, x, y, nt); }, s, reps); }
        RUN(8, 0, "plain") RUN(8, 1, "nt   ") RUN(8, 2, "sc1  ") RUN(8, 3, "sys  ")
        RUN(16, 0, "plain") RUN(16, 1, "nt   ") RUN(16, 2, "sc1  ")
        CK(hipFree(col)); CK(hipFree(val)); CK(hipFree(x)); CK(hipFree(y));
    }
    return 0;
}

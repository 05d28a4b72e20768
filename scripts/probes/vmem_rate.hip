// Micro-probe: per-CU cost of vector-memory instructions by width and pattern, L1/L2-resident data.
// Each workgroup (256 threads) loops over a small per-block buffer (32 KB, L1-resident after the first pass)
// ITER times; reports cycles per wave-instruction (shader clock via wall time x 2.1 GHz is avoided: we time
// with events and print ns per wave-instruction per CU).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int W> struct Vec;            // W = dwords per lane
template <> struct Vec<1> { using T = int; };
template <> struct Vec<2> { using T = int2; };
template <> struct Vec<4> { using T = int4; };
__device__ inline int red(int v) { return v; }
__device__ inline int red(int2 v) { return v.x ^ v.y; }
__device__ inline int red(int4 v) { return v.x ^ v.y ^ v.z ^ v.w; }

// coalesced: lane l reads element (k*256 + tid) of a per-block window of `win` elements
template <int W>
__global__ void __launch_bounds__(256) k_coalesced(const int *buf, int *out, int win_elems, int iters)
{
    using T = typename Vec<W>::T;
    const T *p = reinterpret_cast<const T *>(buf) + (size_t)blockIdx.x * win_elems;
    int acc = 0;
    for (int it = 0; it < iters; it++)
#pragma unroll 8
        for (int k = threadIdx.x; k < win_elems; k += 256) acc ^= red(p[k]);
    if (acc == 0x12345678) out[0] = acc;
}
// gather: lane reads p[idx[..]] where idx is a per-lane pseudo-random index inside `span` elements
template <int W>
__global__ void __launch_bounds__(256) k_gather(const int *buf, const int *idx, int *out, int span, int n_idx, int iters)
{
    using T = typename Vec<W>::T;
    const T *p = reinterpret_cast<const T *>(buf) + (size_t)blockIdx.x * span;
    int acc = 0;
    for (int it = 0; it < iters; it++)
#pragma unroll 8
        for (int k = threadIdx.x; k < n_idx; k += 256) acc ^= red(p[idx[k] % span]);
    if (acc == 0x12345678) out[0] = acc;
}

int main()
{
    const int blocks = 256 * 2;           // 2 workgroups per CU
    const size_t bytes = (size_t)blocks * 32768 * 4;
    int *buf, *idx, *out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&idx, 8192 * 4)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 1, bytes));
    int hidx[8192]; unsigned s = 12345; for (int i = 0; i < 8192; i++) { s = s * 1664525u + 1013904223u; hidx[i] = (s >> 8) & 0xFFFFF; }
    CK(hipMemcpy(idx, hidx, sizeof hidx, hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    const int iters = 200;
#define TIME(NAME, INSTR_PER_WAVE, ...) { auto fn = [&]() { __VA_ARGS__; }; fn(); CK(hipDeviceSynchronize()); CK(hipEventRecord(a)); fn(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b)); \
        float ms; CK(hipEventElapsedTime(&ms, a, b)); double per_cu_instr = 2.0 * 4 * (double)(INSTR_PER_WAVE) * iters; \
        printf("%-44s %8.1f us  %7.2f ns per wave-instr per CU (%.1f clk @2.1GHz)  %6.1f B/clk/CU\n", NAME, ms * 1e3, ms * 1e6 / per_cu_instr, ms * 1e6 / per_cu_instr * 2.1, \
               (double)(BYTES_PER_INSTR) / (ms * 1e6 / per_cu_instr * 2.1)); }
    { const int BYTES_PER_INSTR = 256;  const int win = 8192;  TIME("coalesced dword   (32 KB window/WG)", win / 256, k_coalesced<1><<<blocks, 256>>>(buf, out, win, iters)) }
    { const int BYTES_PER_INSTR = 512;  const int win = 4096;  TIME("coalesced dwordx2 (32 KB window/WG)", win / 256, k_coalesced<2><<<blocks, 256>>>(buf, out, win, iters)) }
    { const int BYTES_PER_INSTR = 1024; const int win = 2048;  TIME("coalesced dwordx4 (32 KB window/WG)", win / 256, k_coalesced<4><<<blocks, 256>>>(buf, out, win, iters)) }
    for (int span : {16, 512, 4096}) {
        char name[96];
        { const int BYTES_PER_INSTR = 256; snprintf(name, 96, "gather dword   span=%d elems", span); TIME(name, 2048 / 256, k_gather<1><<<blocks, 256>>>(buf, idx, out, span, 2048, iters)) }
        { const int BYTES_PER_INSTR = 512; snprintf(name, 96, "gather dwordx2 span=%d elems", span); TIME(name, 2048 / 256, k_gather<2><<<blocks, 256>>>(buf, idx, out, span, 2048, iters)) }
    }
    return 0;
}

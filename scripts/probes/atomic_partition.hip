// Micro-probe (round 3): do u32 atomic adds on random words get cheaper when every XCD only touches its own part of
// the array?  (k_col_count: 8.4 M adds over a 64-MB array take 0.83 ms on R-MAT 24.)
//   all      : every workgroup adds anywhere in the 16.7 M words
//   by-xcd   : workgroup b adds inside part b % 8 (the observed workgroup -> XCD placement)
//   by-group : workgroup b adds inside part b / (grid / 8) (eight XCDs share every part)
// for parts of 8 MB (the whole array split in eight) and of 1 MB.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ unsigned mix(unsigned a)
{
    a ^= a >> 16; a *= 0x7feb352du; a ^= a >> 15; a *= 0x846ca68bu; a ^= a >> 16;
    return a;
}

__global__ void __launch_bounds__(256) k(unsigned *__restrict__ cnt, int iters, int mode, unsigned part_words, unsigned part_stride)
{
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    const unsigned part = mode == 1 ? blockIdx.x % 8 : (mode == 2 ? blockIdx.x / (gridDim.x / 8) : 0);
    for (int it = 0; it < iters; it++) {
        const unsigned h = mix(gid * 0x9E3779B9u + it);
        const unsigned w = mode == 0 ? h & (8 * part_stride - 1) : part * part_stride + (h & (part_words - 1));
        atomicAdd(&cnt[w], 1u);
    }
}

int main()
{
    const unsigned words = 1u << 24;
    unsigned *cnt;
    CK(hipMalloc(&cnt, (size_t)words * 4));
    CK(hipMemset(cnt, 0, (size_t)words * 4));
    const int grid = 2048, iters = 16; // 8.4 M adds
    const char *names[3] = {"all     ", "by-xcd  ", "by-group"};
    for (unsigned part_words : {1u << 21, 1u << 18})
        for (int mode = 0; mode < 3; mode++) {
            hipEvent_t a, b;
            CK(hipEventCreate(&a));
            CK(hipEventCreate(&b));
            float best = 1e30f;
            for (int r = 0; r < 4; r++) {
                CK(hipEventRecord(a));
                hipLaunchKernelGGL(k, dim3(grid), dim3(256), 0, 0, cnt, iters, mode, part_words, 1u << 21);
                CK(hipEventRecord(b));
                CK(hipEventSynchronize(b));
                float ms;
                CK(hipEventElapsedTime(&ms, a, b));
                best = ms < best ? ms : best;
            }
            printf("%s parts of %4u KB: %8.1f us for %.1f M adds (%.2f per ns)\n", names[mode], part_words / 256, best * 1e3,
                   grid * 256.0 * iters / 1e6, grid * 256.0 * iters / (best * 1e6));
        }
    return 0;
}

// Micro-probe: what does a 64-lane x gather cost in the vector-memory path of one CU when a share of the lanes is
// served elsewhere (LDS table)?  Three ways to keep those lanes out of memory:
//   OOB   : every lane issues, table lanes carry the buffer offset 0xFFFFFFFF (range check returns 0)   [k_spmv_hot today]
//   EXEC  : table lanes are switched off in the exec mask for the load
//   ALL   : every lane gathers from memory (no table)
// Persistent shape of k_spmv_hot: 256 workgroups x 1024 threads, each wave runs `iters` rounds of 8 gathers from an
// x of `xbytes` (random columns; L2-resident when small).  Prints ns per gather instruction per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(1024) k(const double *__restrict__ x, const unsigned *__restrict__ cols, int n, int iters,
                                          double *__restrict__ out)
{
    const auto xbuf = __builtin_amdgcn_make_buffer_rsrc(const_cast<double *>(x), (short)0, n * 8, 0x00020000);
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6);
    const unsigned *c = cols + wave * 64 * 8 * 4 + lane; // 4 different column sets, reused round-robin
    unsigned long long acc = 0;
    for (int it = 0; it < iters; it++) {
        const unsigned *ci = c + (it & 3) * 512;
        unsigned cw[8];
#pragma unroll
        for (int i = 0; i < 8; i++)
            cw[i] = ci[i * 64];
        unsigned long long g[8];
#pragma unroll
        for (int i = 0; i < 8; i++) {
            const bool table = cw[i] >> 31;
            if (MODE == 0) { // OOB
                const unsigned off = table ? 0xFFFFFFFFu : cw[i] * 8u;
                g[i] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(xbuf, off, 0, 0));
            } else if (MODE == 1) { // EXEC
                g[i] = 0;
                if (!table)
                    g[i] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(xbuf, cw[i] * 8u, 0, 0));
            } else { // ALL
                g[i] = __builtin_bit_cast(unsigned long long, __builtin_amdgcn_raw_buffer_load_b64(xbuf, (cw[i] & 0x7FFFFFFFu) * 8u, 0, 0));
            }
        }
#pragma unroll
        for (int i = 0; i < 8; i++)
            acc += g[i];
    }
    if (acc == 0x1234567ull)
        out[wave * 64 + lane] = (double)acc;
}

int main(int argc, char **argv)
{
    const int iters = 200, waves = 256 * 16;
    hipStream_t s; CK(hipStreamCreate(&s));
    double *out; CK(hipMalloc(&out, (size_t)waves * 64 * 8));
    for (size_t xbytes : {(size_t)1 << 20, (size_t)8 << 20, (size_t)128 << 20}) {
        const int n = (int)(xbytes / 8);
        double *x; CK(hipMalloc(&x, xbytes)); CK(hipMemset(x, 0, xbytes));
        for (int pct : {0, 34, 50, 66, 90, 100}) { // share of table lanes
            std::vector<unsigned> hc((size_t)waves * 64 * 8 * 4);
            unsigned long long st = 88172645463325252ull;
            auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
            for (auto &v : hc) {
                const unsigned col = (unsigned)(rnd() % n);
                v = (int)(rnd() % 100) < pct ? (0x80000000u | col) : col;
            }
            unsigned *cols; CK(hipMalloc(&cols, hc.size() * 4));
            CK(hipMemcpy(cols, hc.data(), hc.size() * 4, hipMemcpyHostToDevice));
            float ms[3];
            for (int mode = 0; mode < 3; mode++) {
                auto launch = [&]() {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(1024), 0, s, x, cols, n, iters, out);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(1024), 0, s, x, cols, n, iters, out);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(1024), 0, s, x, cols, n, iters, out);
                };
                launch(); CK(hipStreamSynchronize(s));
                hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
                CK(hipEventRecord(a, s)); for (int r = 0; r < 5; r++) launch(); CK(hipEventRecord(b, s)); CK(hipEventSynchronize(b));
                CK(hipEventElapsedTime(&ms[mode], a, b)); ms[mode] /= 5;
            }
            // gather instructions per CU = 16 waves * iters * 8
            const double per = 1e6 / (16.0 * iters * 8);
            printf("x %4zu MB  table lanes %3d%%   OOB %7.1f ns/gather-instr/CU   EXEC %7.1f   ALL %7.1f   (kernel %.0f / %.0f / %.0f us)\n",
                   xbytes >> 20, pct, ms[0] * per, ms[1] * per, ms[2] * per, ms[0] * 1e3, ms[1] * 1e3, ms[2] * 1e3);
            CK(hipFree(cols));
        }
        CK(hipFree(x));
    }
    return 0;
}

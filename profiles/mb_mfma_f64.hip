// microbenchmark: v_mfma_f64_16x16x4_f64 issue rate, fp64 VALU rate, and the two together on one SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double f64x4 __attribute__((ext_vector_type(4)));
// mode 0: all waves MFMA; 1: all waves VALU fma; 2: even waves MFMA, odd waves VALU
__global__ __launch_bounds__(1024) void k(double *out, unsigned long long *cyc, int iters, int mode)
{
    const int wave = threadIdx.x >> 6;
    const bool do_mfma = mode == 0 || (mode == 2 && (wave & 4) == 0);   // waves 0-3 land on SIMDs 0-3, 4-7 again on 0-3, ...
    double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
    f64x4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    double v0 = a, v1 = b, v2 = a + b, v3 = a - b, v4 = a * 2, v5 = b * 2, v6 = a * 3, v7 = b * 3;
    __syncthreads();
    unsigned long long t0 = __builtin_readcyclecounter();
    if (do_mfma) {
        for (int i = 0; i < iters; i++) {
            c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
            c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
        }
    } else {
        for (int i = 0; i < iters; i++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                v0 = __builtin_fma(v0, a, b); v1 = __builtin_fma(v1, a, b); v2 = __builtin_fma(v2, a, b); v3 = __builtin_fma(v3, a, b);
                v4 = __builtin_fma(v4, a, b); v5 = __builtin_fma(v5, a, b); v6 = __builtin_fma(v6, a, b); v7 = __builtin_fma(v7, a, b);
            }
        }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3] + v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x / 64) + wave] = t1 - t0;
}
int main()
{
    const int iters = 2000;
    double *out; unsigned long long *cyc;
    hipMalloc(&out, 1024 * 256 * sizeof(double)); hipMalloc(&cyc, 16 * 256 * sizeof(unsigned long long));
    for (int threads : {256, 512}) {   // 1 or 2 waves per SIMD
        for (int mode = 0; mode < 3; mode++) {
            if (mode == 2 && threads == 256) continue;
            hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, iters, mode);
            hipDeviceSynchronize();
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            hipEventRecord(e0); hipLaunchKernelGGL(k, dim3(256), dim3(threads), 0, 0, out, cyc, iters, mode); hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<unsigned long long> h(threads / 64);
            hipMemcpy(h.data(), cyc, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            printf("threads %d mode %d (%s): %.3f ms; wave cycles (s_memtime ticks):", threads, mode, mode == 0 ? "mfma" : mode == 1 ? "valu" : "waves 0-3 mfma, 4-7 valu", ms);
            for (auto c : h) printf(" %llu", c);
            printf("  -> per mfma %.1f ticks, per 64 fma-instr %.1f ticks\n", (double)h[0] / (4.0 * iters), (double)h.back() / iters);
        }
    }
    return 0;
}

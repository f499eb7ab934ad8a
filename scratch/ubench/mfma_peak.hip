// MFMA-only microbenchmark: v_mfma_i32_16x16x64_i8 and v_mfma_f32_16x16x32_bf16 issue rate on gfx950.
// Every wave keeps NACC independent accumulators and issues back-to-back MFMAs on register operands.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

template <int NACC, bool I8>
__global__ __launch_bounds__(512) void peak(int iters, int *out, unsigned long long *cyc) {
    v4i a = {(int)threadIdx.x, 1, 2, 3}, b = {3, 2, 1, (int)threadIdx.x};
    v4i acc[NACC];
    v4f accf[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) { acc[i] = (v4i){0, 0, 0, 0}; accf[i] = (v4f){0, 0, 0, 0}; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) {
            if constexpr (I8) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
            else accf[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), accf[i], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][3] + (int)accf[i][0];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NACC, bool I8>
static void run(const char *name, int threads, int wgs) {
    int *out; unsigned long long *cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    peak<NACC, I8><<<wgs, threads>>>(100, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    peak<NACC, I8><<<wgs, threads>>>(iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    const double nm = (double)iters * NACC * (threads / 64) * wgs;       // MFMAs
    const double ops = nm * 2.0 * 16 * 16 * (I8 ? 64 : 32);
    printf("%-28s wgs %4d x %3d thr  %8.3f ms  %8.1f Tops/s  wave-cycles/MFMA %.2f  (eff clock %.2f GHz)\n", name, wgs, threads, ms,
           ops / ms * 1e-9, (double)c / ((double)iters * NACC), (double)c / (ms * 1e6));
}
int main() {
    run<24, true>("i8 16x16x64, 1 wave/SIMD", 256, 256);
    run<24, true>("i8 16x16x64, 2 waves/SIMD", 512, 256);
    run<24, true>("i8 16x16x64, 4 waves/SIMD", 512, 512);
    run<4, true>("i8 16x16x64 4 acc 2w/SIMD", 512, 256);
    run<24, false>("bf16 16x16x32, 1 wave/SIMD", 256, 256);
    run<24, false>("bf16 16x16x32, 2 waves/SIMD", 512, 256);
    return 0;
}

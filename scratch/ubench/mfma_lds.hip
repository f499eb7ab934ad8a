// MFMA + LDS-read interaction on gfx950: 24 MFMAs (6 A x 4 B fragments) per iteration, with 0..10 ds_read_b128
// per iteration feeding them.  512-thread workgroups (2 waves/SIMD), one per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
template <int MODE>   // 0: register operands only (distinct regs); 1: A from LDS; 2: A and B from LDS; 3: as 2 with a barrier per iteration
__global__ __launch_bounds__(512) void k(int iters, int *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 65536 / 16; i += 512) ((v4i *)smem)[i] = (v4i){i, i * 3, i * 5, i * 7};
    __syncthreads();
    v4i acc[6][4], a[6], b[4];
    for (int m = 0; m < 6; ++m) { a[m] = (v4i){m, lane, 2, 3}; for (int t = 0; t < 4; ++t) acc[m][t] = (v4i){0, 0, 0, 0}; }
    for (int t = 0; t < 4; ++t) b[t] = (v4i){t, 1, lane, 3};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        const int base = ((it * 7 + wave * 3) & 31) * 1024;
        if constexpr (MODE >= 3) __builtin_amdgcn_s_barrier();
        if constexpr (MODE >= 2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = *(const v4i *)(smem + ((base + 8192 + t * 1024) & 65535) + lane * 16);
        }
#pragma unroll
        for (int m = 0; m < 6; ++m) {
            if constexpr (MODE >= 1) a[m] = *(const v4i *)(smem + ((base + m * 1024) & 65535) + lane * 16);
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[m][t] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], b[t], acc[m][t], 0, 0, 0);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int m = 0; m < 6; ++m) for (int t = 0; t < 4; ++t) s += acc[m][t][0] + acc[m][t][2];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == 448 && blockIdx.x == 0) cyc[0] = t1 - t0;     // a wave of the younger half
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
template <int MODE>
static void run(const char *name) {
    int *out; unsigned long long *cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 16);
    const int iters = 20000;
    hipFuncSetAttribute((const void *)k<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<256, 512, 65536>>>(100, out, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<256, 512, 65536>>>(iters, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 24 * 8 * 256;
    printf("%-44s %8.3f ms %8.1f Tops/s  cycles/iteration: wave7 %.0f wave0 %.0f (ideal 2 waves x 24 x 16 = 768)\n", name, ms,
           nm * 2.0 * 16 * 16 * 64 / ms * 1e-9, (double)c[0] / iters, (double)c[1] / iters);
}
int main() {
    run<0>("registers only");
    run<1>("A from LDS (6 ds_read_b128 / 24 MFMA)");
    run<2>("A and B from LDS (10 / 24)");
    run<3>("A and B from LDS + s_barrier per iteration");
    return 0;
}

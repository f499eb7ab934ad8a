// Does the LDS-read + MFMA serialisation seen in mfma_lds.hip depend on where the accumulators live?
// 24 MFMAs + 10 ds_read_b128 per iteration, compiler-free inner block (inline asm), VGPR vs AGPR accumulators.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rds(v4i &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
template <bool AG> __device__ __forceinline__ void mf(v4i &c, const v4i &a, const v4i &b) {
    if constexpr (AG) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    else asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b));
}
template <bool AG, int NREAD, bool BAR, int THREADS>
__global__ __launch_bounds__(THREADS) void k(int iters, int *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 65536 / 16; i += THREADS) ((v4i *)smem)[i] = (v4i){i, i * 3, i * 5, i * 7};
    __syncthreads();
    v4i acc[6][4], a[6], b[4], n[10];
    for (int m = 0; m < 6; ++m) { a[m] = (v4i){m, lane, 2, 3}; for (int t = 0; t < 4; ++t) acc[m][t] = (v4i){0, 0, 0, 0}; }
    for (int t = 0; t < 4; ++t) b[t] = (v4i){t, 1, lane, 3};
    for (int i = 0; i < 10; ++i) n[i] = a[i % 6];
    const unsigned la = lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        unsigned base = ((it * 7 + wave * 3) & 31) * 1024 + la;
        if constexpr (BAR) __builtin_amdgcn_s_barrier();
        // operands of this iteration were read during the previous one (n[] -> a[], b[])
#pragma unroll
        for (int m = 0; m < 6; ++m) a[m] = n[m];
#pragma unroll
        for (int t = 0; t < 4; ++t) b[t] = n[6 + t];
#pragma unroll
        for (int m = 0; m < 6; ++m) {
#pragma unroll
            for (int t = 0; t < 4; ++t) mf<AG>(acc[m][t], a[m], b[t]);
            // reads spread between the MFMA groups
            if (2 * m < NREAD) rds(n[2 * m], (base + (2 * m) * 1024) & 65535);
            if (2 * m + 1 < NREAD) rds(n[2 * m + 1], (base + (2 * m + 1) * 1024) & 65535);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]), "+v"(n[8]), "+v"(n[9]));
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int m = 0; m < 6; ++m) for (int t = 0; t < 4; ++t) s += acc[m][t][0] + acc[m][t][2];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == THREADS - 64 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
template <bool AG, int NREAD, bool BAR, int THREADS>
static void run(const char *name) {
    int *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 16);
    const int iters = 20000;
    (void)hipFuncSetAttribute((const void *)k<AG, NREAD, BAR, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<AG, NREAD, BAR, THREADS><<<256, THREADS, 65536>>>(100, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<AG, NREAD, BAR, THREADS><<<256, THREADS, 65536>>>(iters, out, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    const double nm = (double)iters * 24 * (THREADS / 64) * 256;
    printf("%-60s %8.3f ms %8.1f Tops/s  cycles/iteration: last wave %.0f wave0 %.0f\n", name, ms,
           nm * 2.0 * 16 * 16 * 64 / ms * 1e-9, (double)c[0] / iters, (double)c[1] / iters);
}
int main() {
    run<false, 0, false, 512>("VGPR acc,  0 reads, 2 waves/SIMD");
    run<false, 10, false, 512>("VGPR acc, 10 reads, 2 waves/SIMD");
    run<true, 0, false, 512>("AGPR acc,  0 reads, 2 waves/SIMD");
    run<true, 10, false, 512>("AGPR acc, 10 reads, 2 waves/SIMD");
    run<true, 10, true, 512>("AGPR acc, 10 reads, 2 waves/SIMD, barrier");
    run<false, 10, true, 512>("VGPR acc, 10 reads, 2 waves/SIMD, barrier");
    run<false, 0, false, 256>("VGPR acc,  0 reads, 1 wave/SIMD");
    run<false, 10, false, 256>("VGPR acc, 10 reads, 1 wave/SIMD");
    run<true, 10, false, 256>("AGPR acc, 10 reads, 1 wave/SIMD");
    return 0;
}

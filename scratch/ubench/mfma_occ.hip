// Occupancy form of the k-step: the same 192 MFMAs per CU and step (48 per SIMD) spread over 8 waves (6 x 4 tiles, 10 reads
// each) or 16 waves (3 x 4 tiles, 7 reads each); operands read one iteration ahead, one read after each of the first MFMAs,
// one barrier per iteration, optional LDS-DMA piece per wave (8 waves) or per second wave (16 waves).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rds(v4i &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void mf(v4i &c, const v4i &a, const v4i &b) { asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
template <int MT, int THREADS, bool DMA>
__global__ __launch_bounds__(THREADS) void k(int iters, const char *src, int *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NR = MT + 4, NM = MT * 4, NWAVE = THREADS / 64;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 98304 / 16; i += THREADS) ((v4i *)smem)[i] = (v4i){i, i * 3, i * 5, i * 7};
    __syncthreads();
    v4i x[NR], y[NR], acc[NM];
    for (int i = 0; i < NR; ++i) { x[i] = (v4i){i, lane, 2, 3}; y[i] = (v4i){lane, i, 1, 3}; }
    for (int i = 0; i < NM; ++i) acc[i] = (v4i){0, 0, 0, 0};
    const unsigned la = lane * 16;
    const char *gsrc = src + ((size_t)blockIdx.x * 16 + wave) * 65536 + lane * 16;
    const bool issuer = DMA && (NWAVE == 8 || (wave & 1) == 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned base = (((it + h) * 7 + wave * 3) & 31) * 1024 + la;
            asm volatile("" : "+v"(base));
            v4i *u = h ? y : x, *n = h ? x : y;
            if (issuer) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            __builtin_amdgcn_s_barrier();
#pragma unroll
            for (int i = 0; i < NM; ++i) {
                mf(acc[i], u[i / 4], u[MT + (i % 4)]);
                if (i < NR) rds(n[i], (base + i * 1024) & 65535);
                if (i == NM / 2 && issuer) {
                    const int slot = ((it + h) * 8 + (wave >> (NWAVE == 16 ? 1 : 0))) & 31;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + (size_t)((it + h) & 63) * 1024),
                                                     (__attribute__((address_space(3))) void *)(smem + 65536 + slot * 1024), 16, 0, 0);
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < NR; ++i) asm volatile("" : "+v"(n[i]));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < NM; ++i) s += acc[i][0] + acc[i][2];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == THREADS - 64 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
template <int MT, int THREADS, bool DMA>
static void run(const char *name, const char *src) {
    int *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 16);
    const int iters = 20000;
    (void)hipFuncSetAttribute((const void *)k<MT, THREADS, DMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MT, THREADS, DMA><<<256, THREADS, 98304>>>(100, src, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MT, THREADS, DMA><<<256, THREADS, 98304>>>(iters, src, out, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    const double macs = (double)iters * MT * 4 * 16384 * (THREADS / 64) * 256;
    printf("%-60s %8.3f ms %8.1f Tops/s  cycles/iteration: last wave %.0f wave0 %.0f (192 MFMAs per CU: 768)\n", name, ms, macs * 2.0 / ms * 1e-9, (double)c[0] / iters, (double)c[1] / iters);
}
int main() {
    char *src; (void)hipMalloc(&src, (size_t)256 * 16 * 65536); (void)hipMemset(src, 1, (size_t)256 * 16 * 65536);
    run<6, 512, false>("8 waves x (6x4 tiles, 10 reads)", src);
    run<3, 1024, false>("16 waves x (3x4 tiles, 7 reads)", src);
    run<6, 512, true>("8 waves x (6x4 tiles, 10 reads) + 8 DMA pieces", src);
    run<3, 1024, true>("16 waves x (3x4 tiles, 7 reads) + 8 DMA pieces", src);
    return 0;
}

// Cost of LDS-DMA issue (global_load_lds_dwordx4) beside MFMAs and ds_reads, 2 waves/SIMD, barrier per iteration.
// Per iteration and wave: 24 MFMAs, 10 ds_read_b128 (two after every 4th MFMA, operands one iteration ahead),
// NDMA global_load_lds_dwordx4 (1 KiB each) into a ring in LDS, counted vmcnt so that PF iterations stay in flight.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rds(v4i &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void mf(v4i &c, const v4i &a, const v4i &b) { asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
#define WAITALL(n) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]), "+v"(n[8]), "+v"(n[9]))
// NDMA DMAs per wave per iteration; WHO: 0 every wave issues its own, 1 only waves 0..3 issue (2x as many), 2 only wave 0 and 4 issue (4x)
template <int NDMA, int WHO, int PLACE, int BARK>
__global__ __launch_bounds__(512) void k(int iters, const char *src, int *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 98304 / 16; i += 512) ((v4i *)smem)[i] = (v4i){i, i * 3, i * 5, i * 7};
    __syncthreads();
    v4i x[10], y[10], acc[24];
    for (int i = 0; i < 10; ++i) { x[i] = (v4i){i, lane, 2, 3}; y[i] = (v4i){lane, i, 1, 3}; }
    for (int i = 0; i < 24; ++i) acc[i] = (v4i){0, 0, 0, 0};
    const unsigned la = lane * 16;
    const char *gsrc = src + ((size_t)blockIdx.x * 8 + wave) * 65536 + lane * 16;
    constexpr int MULT = WHO == 0 ? 1 : (WHO == 1 ? 2 : 4);
    const bool issuer = WHO == 0 ? true : (WHO == 1 ? wave < 4 : (wave & 3) == 0);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned base = (((it + h) * 7 + wave * 3) & 31) * 1024 + la;
            asm volatile("" : "+v"(base));
            v4i *u = h ? y : x, *n = h ? x : y;
            if (BARK == 1 || h == 0) {
                if (NDMA > 0) {
                    if (issuer) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * MULT * 4) : "memory");   // 4 iterations in flight
                }
                __builtin_amdgcn_s_barrier();
            }
            if (PLACE == 0 && NDMA > 0 && issuer) {
#pragma unroll
                for (int d = 0; d < NDMA * MULT; ++d) {
                    const int slot = ((it + h) * NDMA * MULT + d) & 31;
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + (size_t)(((it + h) * 8 + d) & 63) * 1024),
                                                     (__attribute__((address_space(3))) void *)(smem + 65536 + slot * 1024), 16, 0, 0);
                }
            }
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                mf(acc[i], u[i / 4], u[6 + (i % 4)]);
                if ((i % 4) == 3) {
                    const int r = (i / 4) * 2;
                    if (r < 10) rds(n[r], (base + r * 1024) & 65535);
                    if (r + 1 < 10) rds(n[r + 1], (base + (r + 1) * 1024) & 65535);
                }
                if (PLACE == 1 && i == 11 && NDMA > 0 && issuer) {
#pragma unroll
                    for (int d = 0; d < NDMA * MULT; ++d) {
                        const int slot = ((it + h) * NDMA * MULT + d) & 31;
                        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + (size_t)(((it + h) * 8 + d) & 63) * 1024),
                                                         (__attribute__((address_space(3))) void *)(smem + 65536 + slot * 1024), 16, 0, 0);
                    }
                }
            }
            WAITALL(n);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][2];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == 448 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
template <int NDMA, int WHO, int PLACE, int BARK>
static void run(const char *name, const char *src) {
    int *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 16);
    const int iters = 20000;
    (void)hipFuncSetAttribute((const void *)k<NDMA, WHO, PLACE, BARK>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NDMA, WHO, PLACE, BARK><<<256, 512, 98304>>>(100, src, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NDMA, WHO, PLACE, BARK><<<256, 512, 98304>>>(iters, src, out, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    const double macs = (double)iters * 24 * 16384 * 8 * 256;
    printf("%-66s %8.3f ms %8.1f Tops/s  cycles/iteration: wave7 %.0f wave0 %.0f\n", name, ms, macs * 2.0 / ms * 1e-9, (double)c[0] / iters, (double)c[1] / iters);
}
int main() {
    char *src; (void)hipMalloc(&src, (size_t)256 * 8 * 65536); (void)hipMemset(src, 1, (size_t)256 * 8 * 65536);
    run<0, 0, 0, 1>("no DMA, barrier every iteration", src);
    run<0, 0, 0, 2>("no DMA, barrier every 2nd iteration", src);
    run<1, 0, 1, 1>("1 DMA per wave per iteration (mid-block), barrier every iteration", src);
    run<1, 0, 1, 2>("1 DMA per wave per iteration (mid-block), barrier every 2nd iteration", src);
    return 0;
}

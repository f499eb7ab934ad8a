// ds_read_b128 cost beside MFMAs, no register copies: two operand sets, the loop body is unrolled by two.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rds(v4i &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void mf(v4i &c, const v4i &a, const v4i &b) { asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
#define WAITALL(n) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]), "+v"(n[8]), "+v"(n[9]))
// PLACE 0: no reads; 1: one read after each of the first NREAD MFMAs; 2: all reads before the first MFMA; 3: two reads after every 4th MFMA
template <int NREAD, int PLACE, bool BAR, int THREADS>
__global__ __launch_bounds__(THREADS) void k(int iters, int *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 65536 / 16; i += THREADS) ((v4i *)smem)[i] = (v4i){i, i * 3, i * 5, i * 7};
    __syncthreads();
    v4i x[10], y[10], acc[24];
    for (int i = 0; i < 10; ++i) { x[i] = (v4i){i, lane, 2, 3}; y[i] = (v4i){lane, i, 1, 3}; }
    for (int i = 0; i < 24; ++i) acc[i] = (v4i){0, 0, 0, 0};
    const unsigned la = lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned base = (((it + h) * 7 + wave * 3) & 31) * 1024 + la;
            asm volatile("" : "+v"(base));
            v4i *u = h ? y : x, *n = h ? x : y;      // use u (a = u[0..5], b = u[6..9]); read the next operands into n
            if constexpr (BAR) __builtin_amdgcn_s_barrier();
            if (PLACE == 2) {
#pragma unroll
                for (int i = 0; i < NREAD; ++i) rds(n[i], (base + i * 1024) & 65535);
            }
#pragma unroll
            for (int i = 0; i < 24; ++i) {
                mf(acc[i], u[i / 4], u[6 + (i % 4)]);
                if (PLACE == 1 && i < NREAD) rds(n[i], (base + i * 1024) & 65535);
                if (PLACE == 3 && (i % 4) == 3) {
                    const int r = (i / 4) * 2;
                    if (r < NREAD) rds(n[r], (base + r * 1024) & 65535);
                    if (r + 1 < NREAD) rds(n[r + 1], (base + (r + 1) * 1024) & 65535);
                }
            }
            if (NREAD > 0) WAITALL(n);
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][2];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == THREADS - 64 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
template <int NREAD, int PLACE, bool BAR, int THREADS>
static void run(const char *name) {
    int *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 16);
    const int iters = 20000;
    (void)hipFuncSetAttribute((const void *)k<NREAD, PLACE, BAR, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<NREAD, PLACE, BAR, THREADS><<<256, THREADS, 65536>>>(100, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<NREAD, PLACE, BAR, THREADS><<<256, THREADS, 65536>>>(iters, out, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    const double macs = (double)iters * 24 * 16384 * (THREADS / 64) * 256;
    printf("%-58s %8.3f ms %8.1f Tops/s  cycles/iteration: last wave %.0f wave0 %.0f\n", name, ms, macs * 2.0 / ms * 1e-9, (double)c[0] / iters, (double)c[1] / iters);
}
int main() {
    run<0, 0, false, 256>("1 wave/SIMD, no reads");
    run<10, 1, false, 256>("1 wave/SIMD, 10 reads, one after each MFMA");
    run<10, 2, false, 256>("1 wave/SIMD, 10 reads, all first");
    run<10, 3, false, 256>("1 wave/SIMD, 10 reads, two after every 4th MFMA");
    run<0, 0, false, 512>("2 waves/SIMD, no reads");
    run<10, 1, false, 512>("2 waves/SIMD, 10 reads, one after each MFMA");
    run<10, 2, false, 512>("2 waves/SIMD, 10 reads, all first");
    run<10, 3, false, 512>("2 waves/SIMD, 10 reads, two after every 4th MFMA");
    run<0, 0, true, 512>("2 waves/SIMD, no reads, barrier");
    run<10, 1, true, 512>("2 waves/SIMD, 10 reads, one after each MFMA, barrier");
    run<10, 2, true, 512>("2 waves/SIMD, 10 reads, all first, barrier");
    run<10, 3, true, 512>("2 waves/SIMD, 10 reads, two after every 4th, barrier");
    return 0;
}

// ds_read cost beside 8-pass (32x32x32 i8) vs 4-pass (16x16x64 i8) MFMAs, and b64 vs b128 reads.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
typedef int v16i __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void rds(v4i &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void rds64(v2i &d, unsigned addr) { asm volatile("ds_read_b64 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
// SHAPE 0: 24 x 16x16x64; SHAPE 1: 12 x 32x32x32.  RD: 0 none, 1: 10 b128 (one after each MFMA), 2: 20 b64
template <int SHAPE, int RD, int THREADS>
__global__ __launch_bounds__(THREADS) void k(int iters, int *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 65536 / 16; i += THREADS) ((v4i *)smem)[i] = (v4i){i, i * 3, i * 5, i * 7};
    __syncthreads();
    v4i a[6], b[4], n[10];
    v2i n2[20];
    v4i acc4[24];
    v16i acc16[6];
    for (int m = 0; m < 6; ++m) a[m] = (v4i){m, lane, 2, 3};
    for (int t = 0; t < 4; ++t) b[t] = (v4i){t, 1, lane, 3};
    for (int i = 0; i < 10; ++i) n[i] = a[i % 6];
    for (int i = 0; i < 20; ++i) n2[i] = (v2i){i, lane};
    for (int i = 0; i < 24; ++i) acc4[i] = (v4i){0, 0, 0, 0};
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 16; ++j) acc16[i][j] = 0;
    const unsigned la = lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        unsigned base = ((it * 7 + wave * 3) & 31) * 1024 + la;
        if (RD == 1) {
#pragma unroll
            for (int m = 0; m < 6; ++m) a[m] = n[m];
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = n[6 + t];
        }
        if (RD == 2) {
#pragma unroll
            for (int m = 0; m < 6; ++m) a[m] = (v4i){n2[2 * m][0], n2[2 * m][1], n2[2 * m + 1][0], n2[2 * m + 1][1]};
#pragma unroll
            for (int t = 0; t < 4; ++t) b[t] = (v4i){n2[12 + 2 * t][0], n2[12 + 2 * t][1], n2[13 + 2 * t][0], n2[13 + 2 * t][1]};
        }
        constexpr int NM = SHAPE ? 12 : 24;
#pragma unroll
        for (int i = 0; i < NM; ++i) {
            if constexpr (SHAPE == 0) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc4[i]) : "v"(a[i / 4]), "v"(b[i % 4]));
            else asm volatile("v_mfma_i32_32x32x32_i8 %0, %1, %2, %0" : "+v"(acc16[i % 6]) : "v"(a[i % 6]), "v"(b[i % 4]));
            if (RD == 1 && i < 10) rds(n[i], (base + i * 1024) & 65535);
            if (RD == 2) {
                if (SHAPE == 0 && i < 20) rds64(n2[i], (base + i * 512) & 65535);
                if (SHAPE == 1 && i < 10) { rds64(n2[2 * i], (base + i * 1024) & 65535); rds64(n2[2 * i + 1], (base + i * 1024 + 512) & 65535); }
            }
        }
        if (RD == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]), "+v"(n[8]), "+v"(n[9]));
        if (RD == 2) {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n2[0]), "+v"(n2[1]), "+v"(n2[2]), "+v"(n2[3]), "+v"(n2[4]), "+v"(n2[5]), "+v"(n2[6]), "+v"(n2[7]), "+v"(n2[8]), "+v"(n2[9]));
            asm volatile("" : "+v"(n2[10]), "+v"(n2[11]), "+v"(n2[12]), "+v"(n2[13]), "+v"(n2[14]), "+v"(n2[15]), "+v"(n2[16]), "+v"(n2[17]), "+v"(n2[18]), "+v"(n2[19]));
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int i = 0; i < 24; ++i) s += acc4[i][0] + acc4[i][2];
    for (int i = 0; i < 6; ++i) s += acc16[i][0] + acc16[i][9];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == THREADS - 64 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
}
template <int SHAPE, int RD, int THREADS>
static void run(const char *name) {
    int *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 16);
    const int iters = 20000;
    (void)hipFuncSetAttribute((const void *)k<SHAPE, RD, THREADS>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<SHAPE, RD, THREADS><<<256, THREADS, 65536>>>(100, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<SHAPE, RD, THREADS><<<256, THREADS, 65536>>>(iters, out, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    const double macs = (double)iters * 24 * 16384 * (THREADS / 64) * 256;
    printf("%-52s %8.3f ms %8.1f Tops/s  cycles/iteration: last wave %.0f wave0 %.0f\n", name, ms, macs * 2.0 / ms * 1e-9, (double)c[0] / iters, (double)c[1] / iters);
}
int main() {
    run<0, 0, 256>("16x16x64, no reads, 1 wave/SIMD");
    run<0, 1, 256>("16x16x64, 10 b128, 1 wave/SIMD");
    run<0, 2, 256>("16x16x64, 20 b64, 1 wave/SIMD");
    run<1, 0, 256>("32x32x32, no reads, 1 wave/SIMD");
    run<1, 1, 256>("32x32x32, 10 b128, 1 wave/SIMD");
    run<1, 2, 256>("32x32x32, 20 b64, 1 wave/SIMD");
    run<1, 0, 512>("32x32x32, no reads, 2 waves/SIMD");
    run<1, 1, 512>("32x32x32, 10 b128, 2 waves/SIMD");
    run<0, 1, 512>("16x16x64, 10 b128, 2 waves/SIMD");
    return 0;
}

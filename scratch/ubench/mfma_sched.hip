// Schedules for "10 ds_read_b128 + 24 MFMA + 1 barrier per iteration" on gfx950 (2 waves/SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rds(v4i &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
#define WAIT2(n, a, b) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b)::"memory")
#define WAIT4(n, a, b, c, d) asm volatile("s_waitcnt lgkmcnt(" #n ")" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)::"memory")
#define SB __builtin_amdgcn_sched_barrier(0)
#define MF(m, B) \
    acc[m][0] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], B[0], acc[m][0], 0, 0, 0); \
    acc[m][1] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], B[1], acc[m][1], 0, 0, 0); \
    acc[m][2] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], B[2], acc[m][2], 0, 0, 0); \
    acc[m][3] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[m], B[3], acc[m][3], 0, 0, 0);

// MODE 0: kernel-like: a0,a1,B(next) prefetched in the previous iteration; a2..a5 read after the barrier
// MODE 1: as 0, with s_setprio 1 on waves 4..7
// MODE 2: as 0, waves 4..7 staggered by half an iteration (their barrier sits in the middle of their MFMA block)
// MODE 3: a2..a5 of the NEXT iteration read in the second half of this one (everything one iteration ahead)
template <int MODE>
__global__ __launch_bounds__(512) void k(int iters, int *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 65536 / 16; i += 512) ((v4i *)smem)[i] = (v4i){i, i * 3, i * 5, i * 7};
    __syncthreads();
    v4i acc[6][4], a[6], b0[4], b1[4], ap[2];
    for (int m = 0; m < 6; ++m) { a[m] = (v4i){m, lane, 2, 3}; for (int t = 0; t < 4; ++t) acc[m][t] = (v4i){0, 0, 0, 0}; }
    for (int t = 0; t < 4; ++t) { b0[t] = (v4i){t, 1, lane, 3}; b1[t] = b0[t]; }
    ap[0] = a[0]; ap[1] = a[1];
    if (MODE == 1 && wave >= 4) __builtin_amdgcn_s_setprio(1);
    const unsigned la = lane * 16;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    const bool late = (MODE == 2) && wave >= 4;
    for (int it = 0; it < iters; it += 2) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            unsigned base = (((it + h) * 7 + wave * 3) & 31) * 1024 + la;
            asm volatile("" : "+v"(base));
            v4i *bc = h ? b1 : b0, *bn = h ? b0 : b1;
            if constexpr (MODE == 3) {
                __builtin_amdgcn_s_barrier();
                // a[0..5] and bc were read during the previous iteration
                MF(0, bc) MF(1, bc) MF(2, bc)
                SB;
                v4i an[6];
                rds(bn[0], (base + 8192) & 65535); rds(bn[1], (base + 9216) & 65535); rds(bn[2], (base + 10240) & 65535); rds(bn[3], (base + 11264) & 65535);
                rds(an[0], base & 65535); rds(an[1], (base + 1024) & 65535); rds(an[2], (base + 2048) & 65535);
                rds(an[3], (base + 3072) & 65535); rds(an[4], (base + 4096) & 65535); rds(an[5], (base + 5120) & 65535);
                SB;
                MF(3, bc) MF(4, bc) MF(5, bc)
                SB;
                WAIT4(6, bn[0], bn[1], bn[2], bn[3]);
                WAIT2(4, an[0], an[1]); WAIT2(2, an[2], an[3]); WAIT2(0, an[4], an[5]);
#pragma unroll
                for (int m = 0; m < 6; ++m) a[m] = an[m];
                SB;
            } else if (!late) {
                __builtin_amdgcn_s_barrier();
                a[0] = ap[0]; a[1] = ap[1];
                rds(a[2], (base + 2048) & 65535); rds(a[3], (base + 3072) & 65535); rds(a[4], (base + 4096) & 65535); rds(a[5], (base + 5120) & 65535);
                SB;
                MF(0, bc) MF(1, bc)
                SB;
                rds(bn[0], (base + 8192) & 65535); rds(bn[1], (base + 9216) & 65535); rds(bn[2], (base + 10240) & 65535); rds(bn[3], (base + 11264) & 65535);
                rds(ap[0], (base + 6144) & 65535); rds(ap[1], (base + 7168) & 65535);
                SB;
                WAIT2(8, a[2], a[3]);
                MF(2, bc) MF(3, bc)
                SB;
                WAIT2(6, a[4], a[5]);
                MF(4, bc) MF(5, bc)
                SB;
                WAIT4(2, bn[0], bn[1], bn[2], bn[3]);
                WAIT2(0, ap[0], ap[1]);
                SB;
            } else {
                // staggered half: second half of the previous iteration's MFMAs, barrier, first half of this one's
                MF(3, bn) MF(4, bn) MF(5, bn)      // bn holds the PREVIOUS iteration's B here (roles swapped below)
                SB;
                __builtin_amdgcn_s_barrier();
                a[0] = ap[0]; a[1] = ap[1];
                rds(a[2], (base + 2048) & 65535);
                SB;
                MF(0, bc) MF(1, bc)
                SB;
                rds(a[3], (base + 3072) & 65535); rds(a[4], (base + 4096) & 65535); rds(a[5], (base + 5120) & 65535);
                rds(bn[0], (base + 8192) & 65535); rds(bn[1], (base + 9216) & 65535); rds(bn[2], (base + 10240) & 65535); rds(bn[3], (base + 11264) & 65535);
                rds(ap[0], (base + 6144) & 65535); rds(ap[1], (base + 7168) & 65535);
                SB;
                WAIT2(9, a[2], a[2]);
                MF(2, bc)
                SB;
                WAIT2(6, a[3], a[4]); WAIT2(6, a[5], a[5]);
                WAIT4(2, bn[0], bn[1], bn[2], bn[3]);
                WAIT2(0, ap[0], ap[1]);
                SB;
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
    for (int m = 0; m < 6; ++m) for (int t = 0; t < 4; ++t) s += acc[m][t][0] + acc[m][t][2];
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == 448 && blockIdx.x == 0) cyc[0] = t1 - t0;
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
    printf("%-56s %8.3f ms %8.1f Tops/s  cycles/iteration: wave7 %.0f wave0 %.0f\n", name, ms,
           nm * 2.0 * 16 * 16 * 64 / ms * 1e-9, (double)c[0] / iters, (double)c[1] / iters);
}
int main() {
    run<0>("kernel-like manual schedule");
    run<1>("  + s_setprio 1 on waves 4..7");
    run<2>("  + waves 4..7 staggered by half an iteration");
    run<3>("everything read one iteration ahead");
    return 0;
}

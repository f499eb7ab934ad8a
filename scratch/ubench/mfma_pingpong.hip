// Round 4: SIMD partners out of phase on the HARDWARE barrier (MI355X_MICROARCH.md "Two waves per SIMD", item 9).
// The ring kernel's k-step per wave: 24 v_mfma_i32_16x16x64_i8, 10 ds_read_b128, NDMA global_load_lds_dwordx4 (1 KiB each),
// a counted vmcnt wait.  Forms (all 512 threads, one workgroup per CU unless said otherwise):
//   MODE 0  lock-step, one barrier per k-step, reads two behind every 4th MFMA into a second operand set (the production shape)
//   MODE 1  two half-phases per k-step: LOAD (wait, DMAs, the 10 reads, lgkmcnt(0)) | barrier | MFMA (24, s_setprio 1) | barrier,
//           waves 4..7 one barrier behind waves 0..3: one partner computes while the other loads
//   MODE 2  the same with the ODD waves behind (the guide says this pairing is the wrong one)
//   MODE 3  the same two half-phases with nobody behind (both partners load together, then compute together)
//   MODE 5  ONE barrier per k-step; every wave runs [MFMAs of step s][DMAs, reads of step s+1, wait]; waves 4..7 have the barrier in front of
//           the MFMAs, waves 0..3 behind them: between two barriers one partner computes-then-loads, the other loads-then-computes
//   MODE 6  the same split by wave parity
//   MODE 4  one wave per SIMD (256 threads): 48 MFMAs, 14 reads spread over them, 2 * NDMA DMAs, one barrier
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int v4i __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void rds(v4i &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr) : "memory"); }
__device__ __forceinline__ void mf(v4i &c, const v4i &a, const v4i &b) { asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(c) : "v"(a), "v"(b)); }
#define WAITALL10(n) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]), "+v"(n[8]), "+v"(n[9]))
#define WAITALL14(n) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(n[0]), "+v"(n[1]), "+v"(n[2]), "+v"(n[3]), "+v"(n[4]), "+v"(n[5]), "+v"(n[6]), "+v"(n[7]), "+v"(n[8]), "+v"(n[9]), "+v"(n[10]), "+v"(n[11]), "+v"(n[12]), "+v"(n[13]))

template <int NDMA>
__device__ __forceinline__ void dmas(const char *gsrc, char *smem, int it) {
#pragma unroll
    for (int d = 0; d < NDMA; ++d) {
        const int slot = (it * NDMA + d) & 31;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(gsrc + (size_t)((it * 8 + d) & 63) * 1024),
                                         (__attribute__((address_space(3))) void *)(smem + 65536 + slot * 1024), 16, 0, 0);
    }
}

template <int MODE, int NDMA, int PRIO>
__global__ __launch_bounds__(MODE == 4 ? 256 : 512) void k(int iters, const char *src, int *out, unsigned long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int THREADS = MODE == 4 ? 256 : 512;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    for (int i = threadIdx.x; i < 98304 / 16; i += THREADS) ((v4i *)smem)[i] = (v4i){i, i * 3, i * 5, i * 7};
    __syncthreads();
    const unsigned la = lane * 16;
    const char *gsrc = src + ((size_t)(blockIdx.x & 7) * 8 + wave) * 65536 + lane * 16;   // 4 MiB in all: L2-resident, like the weights
    unsigned long long t0 = 0, t1 = 0;
    int s = 0;
    if constexpr (MODE == 0) {
        v4i x[10], y[10], acc[24];
        for (int i = 0; i < 10; ++i) { x[i] = (v4i){i, lane, 2, 3}; y[i] = (v4i){lane, i, 1, 3}; }
        for (int i = 0; i < 24; ++i) acc[i] = (v4i){0, 0, 0, 0};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned base = (((it + h) * 7 + wave * 3) & 31) * 1024 + la;
                asm volatile("" : "+v"(base));
                v4i *u = h ? y : x, *n = h ? x : y;
                if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * 4) : "memory");
                __builtin_amdgcn_s_barrier();
                dmas<NDMA>(gsrc, smem, it + h);
#pragma unroll
                for (int i = 0; i < 24; ++i) {
                    mf(acc[i], u[i / 4], u[6 + (i % 4)]);
                    if ((i % 4) == 3) {
                        const int r = (i / 4) * 2;
                        if (r < 10) rds(n[r], (base + r * 1024) & 65535);
                        if (r + 1 < 10) rds(n[r + 1], (base + (r + 1) * 1024) & 65535);
                    }
                }
                WAITALL10(n);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][2];
    } else if constexpr (MODE >= 1 && MODE <= 3) {
        v4i n[10], acc[24];
        for (int i = 0; i < 10; ++i) n[i] = (v4i){i, lane, 2, 3};
        for (int i = 0; i < 24; ++i) acc[i] = (v4i){0, 0, 0, 0};
        const bool behind = MODE == 1 ? wave >= 4 : (MODE == 2 ? (wave & 1) : false);
        if (PRIO == 2 && wave >= 4) __builtin_amdgcn_s_setprio(1);
        t0 = __builtin_amdgcn_s_memtime();
        if (behind) __builtin_amdgcn_s_barrier();
        for (int it = 0; it < iters; ++it) {
            unsigned base = ((it * 7 + wave * 3) & 31) * 1024 + la;
            asm volatile("" : "+v"(base));
            // LOAD half-phase
            if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * 4) : "memory");
            dmas<NDMA>(gsrc, smem, it);
#pragma unroll
            for (int i = 0; i < 10; ++i) rds(n[i], (base + i * 1024) & 65535);
            WAITALL10(n);
            __builtin_amdgcn_s_barrier();
            // MFMA half-phase
            if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 24; ++i) mf(acc[i], n[i / 4], n[6 + (i % 4)]);
            if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
            __builtin_amdgcn_s_barrier();
        }
        if (!behind && MODE != 3) __builtin_amdgcn_s_barrier();
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][2];
    } else if constexpr (MODE == 5 || MODE == 6) {
        v4i n[10], acc[24];
        for (int i = 0; i < 10; ++i) n[i] = (v4i){i, lane, 2, 3};
        for (int i = 0; i < 24; ++i) acc[i] = (v4i){0, 0, 0, 0};
        const bool grp = MODE == 5 ? wave >= 4 : (wave & 1);
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; ++it) {
            unsigned base = ((it * 7 + wave * 3) & 31) * 1024 + la;
            asm volatile("" : "+v"(base));
            if (grp) __builtin_amdgcn_s_barrier();
            if (PRIO == 1) __builtin_amdgcn_s_setprio(1);
#pragma unroll
            for (int i = 0; i < 24; ++i) mf(acc[i], n[i / 4], n[6 + (i % 4)]);
            if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
            if (!grp) __builtin_amdgcn_s_barrier();
            dmas<NDMA>(gsrc, smem, it);
#pragma unroll
            for (int i = 0; i < 10; ++i) rds(n[i], (base + i * 1024) & 65535);
            if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * 4) : "memory");
            WAITALL10(n);
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 24; ++i) s += acc[i][0] + acc[i][2];
    } else {
        v4i x[14], y[14], acc[48];
        for (int i = 0; i < 14; ++i) { x[i] = (v4i){i, lane, 2, 3}; y[i] = (v4i){lane, i, 1, 3}; }
        for (int i = 0; i < 48; ++i) acc[i] = (v4i){0, 0, 0, 0};
        t0 = __builtin_amdgcn_s_memtime();
        for (int it = 0; it < iters; it += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                unsigned base = (((it + h) * 7 + wave * 3) & 31) * 1024 + la;
                asm volatile("" : "+v"(base));
                v4i *u = h ? y : x, *n = h ? x : y;
                if (NDMA > 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NDMA * 2 * 4) : "memory");
                __builtin_amdgcn_s_barrier();
#pragma unroll
                for (int i = 0; i < 48; ++i) {
                    mf(acc[i], u[i / 8], u[6 + (i % 8)]);
                    if ((i % 3) == 2 && i / 3 < 14) rds(n[i / 3], (base + (i / 3) * 1024) & 65535);
                    if (NDMA > 0 && (i == 20 || i == 44)) dmas<NDMA>(gsrc, smem, (it + h) * 2 + (i > 30));
                }
                WAITALL14(n);
            }
        }
        t1 = __builtin_amdgcn_s_memtime();
        for (int i = 0; i < 48; ++i) s += acc[i][0] + acc[i][2];
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (s == 0x7fffffff) out[0] = s;
    if (threadIdx.x == THREADS - 64 && blockIdx.x == 0) cyc[0] = t1 - t0;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[1] = t1 - t0;
}

template <int MODE, int NDMA, int PRIO>
static void run(const char *name, const char *src) {
    constexpr int THREADS = MODE == 4 ? 256 : 512;
    int *out; unsigned long long *cyc;
    (void)hipMalloc(&out, 4); (void)hipMalloc(&cyc, 16);
    const int iters = 20000;
    (void)hipFuncSetAttribute((const void *)k<MODE, NDMA, PRIO>, hipFuncAttributeMaxDynamicSharedMemorySize, 98304);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    k<MODE, NDMA, PRIO><<<256, THREADS, 98304>>>(100, src, out, cyc);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    k<MODE, NDMA, PRIO><<<256, THREADS, 98304>>>(iters, src, out, cyc);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    unsigned long long c[2]; (void)hipMemcpy(c, cyc, 16, hipMemcpyDeviceToHost);
    const double macs = (double)iters * 24 * 16384 * 8 * 256;     // the same work per k-step in every mode
    printf("%-74s %8.3f ms %8.1f Tops/s  cycles per k-step: last wave %.0f wave0 %.0f\n", name, ms, macs * 2.0 / ms * 1e-9, (double)c[0] / iters,
           (double)c[1] / iters);
    (void)hipFree(out); (void)hipFree(cyc);
}
int main() {
    char *src;
    (void)hipMalloc(&src, (size_t)256 * 8 * 65536);
    (void)hipMemset(src, 1, (size_t)256 * 8 * 65536);
    for (int rep = 0; rep < 2; ++rep) {
        run<0, 0, 0>("lock-step, one barrier, reads interleaved, no DMA", src);
        run<0, 2, 0>("lock-step, one barrier, reads interleaved, 2 DMA", src);
        run<3, 0, 0>("two half-phases, nobody behind, no DMA", src);
        run<3, 2, 0>("two half-phases, nobody behind, 2 DMA", src);
        run<1, 0, 0>("two half-phases, waves 4-7 behind, no DMA", src);
        run<1, 2, 0>("two half-phases, waves 4-7 behind, 2 DMA", src);
        run<1, 2, 1>("two half-phases, waves 4-7 behind, 2 DMA, setprio 1 in the MFMA half", src);
        run<1, 2, 2>("two half-phases, waves 4-7 behind, 2 DMA, static prio 1 for waves 4-7", src);
        run<1, 3, 1>("two half-phases, waves 4-7 behind, 3 DMA, setprio 1 in the MFMA half", src);
        run<2, 2, 1>("two half-phases, ODD waves behind, 2 DMA, setprio 1 in the MFMA half", src);
        run<0, 1, 0>("lock-step, one barrier, reads interleaved, 1 DMA", src);
        run<5, 0, 0>("ONE barrier, waves 4-7 compute-then-load, 0-3 load-then-compute, no DMA", src);
        run<5, 1, 0>("ONE barrier, waves 4-7 compute-then-load, 0-3 load-then-compute, 1 DMA", src);
        run<5, 2, 0>("ONE barrier, waves 4-7 compute-then-load, 0-3 load-then-compute, 2 DMA", src);
        run<5, 2, 1>("ONE barrier, waves 4-7 compute-then-load, 2 DMA, setprio 1 in the MFMA half", src);
        run<6, 2, 0>("ONE barrier, ODD waves compute-then-load, 2 DMA", src);
    }
    return 0;
}

// Round 5: does a SIMD run a matrix-only wave and a vector-only wave CONCURRENTLY?
//
// valu_issue.hip (round 4) only ever ran HOMOGENEOUS waves: every wave carries the same MFMA + VALU mix, and the finding was
// "the costs add".  MI355X_MICROARCH.md says that an MFMA-only wave and a VALU-only wave overlap.  This benchmark gives the
// waves of a SIMD different ROLES at EQUAL WORK PER SIMD and prints cycles per unit of work:
//
//   unit of work = 8 v_mfma_i32_16x16x64_i8  +  64 vector instructions in the production epilogue mix of front.hip / convpx.hip
//                  (per 4 outputs: 4 pool max, 2 fma per output, 1 SDWA max-and-pack per output, max3 / min3 clamp detection;
//                  plus address / pack odds and ends)
//
//   homogeneous  W waves per SIMD, each wave one unit per trip (MFMAs interleaved 1 : 8 with the VALU, or all MFMAs first)
//   roles (a+b)  a matrix-only waves and b vector-only waves per SIMD; a matrix wave issues MU units' MFMAs per trip, a vector
//                wave VU units' VALU, a * MU = b * VU units per SIMD and trip
//   hand-off     none: the roles run side by side (upper bound of what specialisation could buy)
//                raw : matrix waves ds_read_b128 one B fragment per MFMA and ds_write_b128 every accumulator; vector waves
//                      ds_read_b128 them back (what "MFMA waves hand raw accumulators through LDS" costs)
//                pool: matrix waves max-pool four accumulators (8 VALU) and write one; vector waves read one per 4 MFMAs
//   sync         one s_barrier per trip in every wave (the coupling a real producer / consumer pair needs at least)
//   prio         s_setprio 1 on the matrix waves
// A wave sits on SIMD (wave & 3) (HW_REG_HW_ID, profiles/r04_notes.md section 1), so waves w, w + 4, w + 8 ... share a SIMD; the
// matrix waves are the OLDEST of their SIMD unless `young` says otherwise.
// Output: cycles (s_memtime) per unit and SIMD = (slowest wave of the workgroup, median over 256 workgroups) / units.
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef int v4i __attribute__((ext_vector_type(4)));

#define A_MFMA(acc) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc) : "v"(wa), "v"(wb))

// one unit's 64 vector instructions: 2 x [4 v_max3_i32 + 4 v_max_i32 | 8 v_fma_f32 | 4 v_max_f32_sdwa | 2 v_max3_f32 + 2 v_min3_f32] + 16 odds
#define VALU24(o)                                                                                                                  \
    asm volatile("v_max3_i32 %0, %0, %1, %2\n v_max3_i32 %3, %3, %1, %2\n v_max3_i32 %4, %4, %1, %2\n v_max3_i32 %5, %5, %1, %2"   \
                 : "+v"(x[o + 0]), "+v"(ic), "+v"(id), "+v"(x[o + 1]), "+v"(x[o + 2]), "+v"(x[o + 3]));                            \
    asm volatile("v_max_i32 %0, %0, %1\n v_max_i32 %2, %2, %1\n v_max_i32 %3, %3, %1\n v_max_i32 %4, %4, %1"                       \
                 : "+v"(x[o + 0]), "+v"(ic), "+v"(x[o + 1]), "+v"(x[o + 2]), "+v"(x[o + 3]));                                      \
    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"       \
                 : "+v"(f[o + 0]), "+v"(f[o + 1]), "+v"(f[o + 2]), "+v"(f[o + 3]) : "v"(m), "v"(c));                               \
    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"       \
                 : "+v"(h[o + 0]), "+v"(h[o + 1]), "+v"(h[o + 2]), "+v"(h[o + 3]) : "v"(m), "v"(c));                               \
    asm volatile("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"                  \
                 "v_max_f32_sdwa %0, %3, %4 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"             \
                 "v_max_f32_sdwa %0, %5, %6 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"             \
                 "v_max_f32_sdwa %0, %7, %8 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"               \
                 : "+v"(pk[(o) >> 2]) : "v"(f[o + 0]), "v"(h[o + 0]), "v"(f[o + 1]), "v"(h[o + 1]), "v"(f[o + 2]), "v"(h[o + 2]),  \
                   "v"(f[o + 3]), "v"(h[o + 3]));                                                                                  \
    asm volatile("v_max3_f32 %0, %0, %2, %3\n v_max3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %6, %7\n v_min3_f32 %1, %1, %8, %9"   \
                 : "+v"(ymx), "+v"(ymn) : "v"(f[o + 0]), "v"(f[o + 1]), "v"(f[o + 2]), "v"(f[o + 3]), "v"(h[o + 0]), "v"(h[o + 1]),\
                   "v"(h[o + 2]), "v"(h[o + 3]));
#define VALU_ODDS8(o)                                                                                                              \
    asm volatile("v_add_u32 %0, %0, %4\n v_lshl_add_u32 %1, %1, 1, %4\n v_and_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n"             \
                 "v_add_u32 %0, %0, %5\n v_perm_b32 %1, %1, %4, %5\n v_or_b32 %2, %2, %5\n v_mul_f32 %3, %3, %6"                   \
                 : "+v"(x[o + 0]), "+v"(x[o + 1]), "+v"(x[o + 2]), "+v"(x[o + 3]) : "v"(ic), "v"(id), "v"(m));
// the vector part of one unit (64 instructions); POOLED hand-off: the matrix wave did the 8 pool max of each half already
#define VALU_UNIT(POOLDONE)                                                   \
    do {                                                                      \
        if (!(POOLDONE)) { VALU24(0) VALU24(4) }                              \
        else { VALU16(0) VALU16(4) }                                          \
        VALU_ODDS8(0) VALU_ODDS8(4)                                           \
    } while (0)
#define VALU16(o)                                                                                                                  \
    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"       \
                 : "+v"(f[o + 0]), "+v"(f[o + 1]), "+v"(f[o + 2]), "+v"(f[o + 3]) : "v"(m), "v"(c));                               \
    asm volatile("v_fma_f32 %0, %0, %4, %5\n v_fma_f32 %1, %1, %4, %5\n v_fma_f32 %2, %2, %4, %5\n v_fma_f32 %3, %3, %4, %5"       \
                 : "+v"(h[o + 0]), "+v"(h[o + 1]), "+v"(h[o + 2]), "+v"(h[o + 3]) : "v"(m), "v"(c));                               \
    asm volatile("v_max_f32_sdwa %0, %1, %2 dst_sel:BYTE_0 dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:DWORD\n"                  \
                 "v_max_f32_sdwa %0, %3, %4 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"             \
                 "v_max_f32_sdwa %0, %5, %6 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD\n"             \
                 "v_max_f32_sdwa %0, %7, %8 dst_sel:BYTE_3 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD"               \
                 : "+v"(pk[(o) >> 2]) : "v"(f[o + 0]), "v"(h[o + 0]), "v"(f[o + 1]), "v"(h[o + 1]), "v"(f[o + 2]), "v"(h[o + 2]),  \
                   "v"(f[o + 3]), "v"(h[o + 3]));                                                                                  \
    asm volatile("v_max3_f32 %0, %0, %2, %3\n v_max3_f32 %0, %0, %4, %5\n v_min3_f32 %1, %1, %6, %7\n v_min3_f32 %1, %1, %8, %9"   \
                 : "+v"(ymx), "+v"(ymn) : "v"(f[o + 0]), "v"(f[o + 1]), "v"(f[o + 2]), "v"(f[o + 3]), "v"(h[o + 0]), "v"(h[o + 1]),\
                   "v"(h[o + 2]), "v"(h[o + 3]));
// HOMO: 0 roles, 1 homogeneous interleaved, 2 homogeneous MFMAs first.  NM = matrix waves per SIMD, MU / VU units per trip,
// HAND 0 none / 1 raw / 2 pooled, SYNC barrier per trip, PRIO setprio on the matrix waves, YOUNG matrix waves are the youngest
template <int HOMO, int NM, int NWS, int MU, int VU, int HAND, int SYNC, int PRIO, int YOUNG>
__global__ __launch_bounds__(NWS * 256) void k(int iters, float *out, unsigned long long *cyc) {
    __shared__ __attribute__((aligned(16))) int lds[16 * 1024];          // 64 KB: fragments / accumulators, conflict-free lane * 16
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int slot = wave >> 2;                                            // age rank of the wave on its SIMD
    const bool matrix = HOMO ? false : (YOUNG ? slot >= NWS - NM : slot < NM);
    int x[8], ic = lane, id = lane ^ 5;
    float f[8], h[8], ymx = 0.f, ymn = 0.f;
    unsigned int pk[2] = {0u, 0u};
    v4i acc[8], wa = {lane, 1, 2, 3}, wb = {3, lane, 1, 0};
    for (int i = 0; i < 8; ++i) { x[i] = lane + i; f[i] = lane + i; h[i] = lane - i; acc[i] = (v4i){0, 0, 0, 0}; }
    const float m = 1.0001f, c = 0.5f;
    const unsigned int lb = (unsigned int)(unsigned long long)(__attribute__((address_space(3))) int *)lds + lane * 16 + (wave & 3) * 1024;   // this SIMD's 1 KB row
    for (int i = threadIdx.x; i < 16 * 1024; i += blockDim.x) lds[i] = i;
    __syncthreads();
    if (PRIO && matrix) __builtin_amdgcn_s_setprio(1);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    // one loop PER ROLE behind a scalar branch (a role test inside the loop is a divergent branch to the compiler, which then
    // copies every live register of the other role round the loop: ~50 v_mov per trip in the first version of this file)
    auto homo_trip = [&]() {
        if constexpr (HOMO == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                A_MFMA(acc[j]);
                if (j == 0) { VALU24(0) }
                if (j == 2) { VALU24(4) }
                if (j == 4) { VALU_ODDS8(0) }
                if (j == 6) { VALU_ODDS8(4) }
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) A_MFMA(acc[j]);
            VALU_UNIT(false);
        }
    };
    // the accumulators handed over are those of the group of four MFMAs BEFORE the one just issued (64 pipe cycles old: no
    // MFMA-result hazard, as a software-pipelined producer would do it)
    auto hand = [&](int a) {
        if constexpr (HAND == 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) asm volatile("ds_write_b128 %0, %1 offset:4096" ::"v"(lb), "v"(acc[a + q]) : "memory");
        }
        if constexpr (HAND == 2) {
            v4i pl;
#pragma unroll
            for (int r = 0; r < 4; ++r) pl[r] = max(max(acc[a][r], acc[a + 1][r]), max(acc[a + 2][r], acc[a + 3][r]));
            asm volatile("ds_write_b128 %0, %1 offset:4096" ::"v"(lb), "v"(pl) : "memory");
        }
    };
    auto matrix_trip = [&]() {
#pragma unroll
        for (int u = 0; u < MU; ++u) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if constexpr (HAND != 0) {                          // one B fragment per MFMA from LDS, one read ahead
                    v4i t;
                    asm volatile("ds_read_b128 %0, %1" : "=v"(t) : "v"(lb) : "memory");
                    asm volatile("s_waitcnt lgkmcnt(5)" ::: "memory");
                    wb = t;
                }
                A_MFMA(acc[j]);
                if (j == 3) hand(4);
                if (j == 7) hand(0);
            }
        }
    };
    auto vector_trip = [&]() {
#pragma unroll
        for (int u = 0; u < VU; ++u) {
            if constexpr (HAND == 1) {                              // 8 accumulators back
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    v4i t;
                    asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(t) : "v"(lb) : "memory");
                    x[j] ^= t[0];
                }
            }
            if constexpr (HAND == 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    v4i t;
                    asm volatile("ds_read_b128 %0, %1 offset:8192" : "=v"(t) : "v"(lb) : "memory");
                    x[j] ^= t[0];
                }
            }
            VALU_UNIT(HAND == 2);
            if constexpr (HAND != 0) {
                asm volatile("ds_write_b64 %0, %1 offset:12288" ::"v"(lb), "v"(*(unsigned long long *)pk) : "memory");
                asm volatile("s_waitcnt lgkmcnt(1)" ::: "memory");
            }
        }
    };
    if constexpr (HOMO != 0) {
        for (int it = 0; it < iters; ++it) {
            homo_trip();
            if constexpr (SYNC) __builtin_amdgcn_s_barrier();
        }
    } else if (matrix) {
        for (int it = 0; it < iters; ++it) {
            matrix_trip();
            if constexpr (SYNC) __builtin_amdgcn_s_barrier();
        }
    } else {
        for (int it = 0; it < iters; ++it) {
            vector_trip();
            if constexpr (SYNC) __builtin_amdgcn_s_barrier();
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = ymx + ymn + (float)pk[0] + (float)pk[1] + (float)ic + (float)id + (float)wb[0];
    for (int i = 0; i < 8; ++i) r += (float)x[i] + f[i] + h[i] + (float)(acc[i][0] + acc[i][3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (lane == 0) cyc[blockIdx.x * 16 + wave] = t1 - t0;
}

template <int HOMO, int NM, int NWS, int MU, int VU, int HAND, int SYNC, int PRIO, int YOUNG>
static double run(const char *name) {
    const int iters = 1000, grid = 256;
    static float *out = nullptr;
    static unsigned long long *cyc = nullptr;
    if (!out) { hipMalloc(&out, grid * 1024 * 4); hipMalloc(&cyc, grid * 16 * 8); }
    hipMemset(cyc, 0, grid * 16 * 8);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto kk = k<HOMO, NM, NWS, MU, VU, HAND, SYNC, PRIO, YOUNG>;
    hipLaunchKernelGGL(kk, dim3(grid), dim3(NWS * 256), 0, 0, iters, out, cyc);
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(kk, dim3(grid), dim3(NWS * 256), 0, 0, iters, out, cyc);
    hipEventRecord(e1, 0);
    hipError_t err = hipDeviceSynchronize();
    if (err != hipSuccess || hipGetLastError() != hipSuccess) { printf("%-78s launch failed: %s\n", name, hipGetErrorString(err)); return 0; }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> hc(grid * 16);
    hipMemcpy(hc.data(), cyc, grid * 16 * 8, hipMemcpyDeviceToHost);
    std::vector<double> slow(grid), mtx(grid), vec(grid);
    for (int b = 0; b < grid; ++b) {
        unsigned long long mx = 0, mm = 0, mv = 0;
        for (int w = 0; w < NWS * 4; ++w) {
            const unsigned long long v = hc[b * 16 + w];
            mx = std::max(mx, v);
            const int slot = w >> 2;
            const bool matrix = HOMO ? false : (YOUNG ? slot >= NWS - NM : slot < NM);
            if (matrix) mm = std::max(mm, v); else mv = std::max(mv, v);
        }
        slow[b] = (double)mx; mtx[b] = (double)mm; vec[b] = (double)mv;
    }
    std::sort(slow.begin(), slow.end());
    std::sort(mtx.begin(), mtx.end());
    std::sort(vec.begin(), vec.end());
    const double units = HOMO ? NWS : NM ? (double)NM * MU : (double)NWS * VU;                      // per SIMD and trip
    const double per_unit = slow[grid / 2] / iters / units;
    printf("%-78s %7.1f cyc/unit/SIMD   (matrix waves %7.1f, vector waves %7.1f; %6.1f ns/unit wall)\n", name, per_unit,
           mtx[grid / 2] / iters / units, vec[grid / 2] / iters / units, ms * 1e6 / iters / units);
    hipEventDestroy(e0);
    hipEventDestroy(e1);
    return per_unit;
}

int main() {
    printf("unit = 8 MFMA 16x16x64 i8 (128 pipe cycles) + 64 VALU (production epilogue mix); cycles per unit and SIMD, lower is better\n");
    //   HOMO NM NWS MU VU HAND SYNC PRIO YOUNG
    printf("-- homogeneous waves (round 4's regime)\n");
    const double h2 = run<1, 0, 2, 1, 1, 0, 0, 0, 0>("homogeneous, 2 waves/SIMD, MFMA : VALU interleaved");
    const double h4 = run<1, 0, 4, 1, 1, 0, 0, 0, 0>("homogeneous, 4 waves/SIMD, interleaved");
    run<2, 0, 2, 1, 1, 0, 0, 0, 0>("homogeneous, 2 waves/SIMD, MFMAs first");
    run<2, 0, 4, 1, 1, 0, 0, 0, 0>("homogeneous, 4 waves/SIMD, MFMAs first");
    run<1, 0, 4, 1, 1, 0, 1, 0, 0>("homogeneous, 4 waves/SIMD, interleaved, barrier per trip");
    printf("-- the two parts alone (one role present only)\n");
    run<0, 1, 1, 1, 1, 0, 0, 0, 0>("matrix wave alone (1/SIMD): 8 MFMA per unit");
    run<0, 2, 2, 1, 1, 0, 0, 0, 0>("matrix waves alone (2/SIMD)");
    run<0, 0, 1, 1, 1, 0, 0, 0, 0>("vector wave alone (1/SIMD): 64 VALU per unit");
    run<0, 0, 2, 1, 1, 0, 0, 0, 0>("vector waves alone (2/SIMD)");
    run<0, 0, 4, 1, 1, 0, 0, 0, 0>("vector waves alone (4/SIMD)");
    printf("-- roles, no hand-off (upper bound of specialisation)\n");
    const double r11 = run<0, 1, 2, 1, 1, 0, 0, 0, 0>("roles 1 matrix + 1 vector");
    run<0, 1, 2, 1, 1, 0, 0, 1, 0>("roles 1 + 1, setprio 1 on the matrix wave");
    run<0, 1, 2, 1, 1, 0, 0, 0, 1>("roles 1 + 1, matrix wave is the younger");
    const double r22 = run<0, 2, 4, 1, 1, 0, 0, 0, 0>("roles 2 + 2");
    run<0, 2, 4, 1, 1, 0, 0, 1, 0>("roles 2 + 2, setprio 1 on the matrix waves");
    const double r13 = run<0, 1, 4, 3, 1, 0, 0, 0, 0>("roles 1 + 3 (matrix wave 24 MFMA per trip)");
    run<0, 1, 4, 3, 1, 0, 0, 1, 0>("roles 1 + 3, setprio 1 on the matrix wave");
    run<0, 1, 4, 3, 1, 0, 0, 0, 1>("roles 1 + 3, matrix wave is the youngest");
    run<0, 1, 3, 2, 1, 0, 0, 0, 0>("roles 1 + 2");
    printf("-- roles with the LDS hand-off a real producer / consumer pair needs\n");
    run<0, 1, 2, 1, 1, 1, 0, 0, 0>("roles 1 + 1, raw accumulators through LDS");
    run<0, 1, 2, 1, 1, 2, 0, 0, 0>("roles 1 + 1, pooled in the matrix wave, one accumulator in four through LDS");
    run<0, 2, 4, 1, 1, 1, 0, 0, 0>("roles 2 + 2, raw");
    const double r22p = run<0, 2, 4, 1, 1, 2, 0, 0, 0>("roles 2 + 2, pooled");
    run<0, 2, 4, 1, 1, 2, 1, 0, 0>("roles 2 + 2, pooled, barrier per trip");
    run<0, 2, 4, 1, 1, 2, 1, 1, 0>("roles 2 + 2, pooled, barrier per trip, setprio");
    run<0, 1, 4, 3, 1, 1, 0, 0, 0>("roles 1 + 3, raw");
    const double r13p = run<0, 1, 4, 3, 1, 2, 0, 0, 0>("roles 1 + 3, pooled");
    run<0, 1, 4, 3, 1, 2, 1, 0, 0>("roles 1 + 3, pooled, barrier per trip");
    run<0, 1, 4, 3, 1, 2, 1, 1, 0>("roles 1 + 3, pooled, barrier per trip, setprio");
    printf("-- decision (VERDICT r4: go on to a producer / consumer front end if roles >= 1.3 x homogeneous at equal work)\n");
    printf("best homogeneous %.1f (2 w) / %.1f (4 w); roles without hand-off: 1+1 %.1f (x%.2f), 2+2 %.1f (x%.2f), 1+3 %.1f (x%.2f); "
           "with pooled hand-off: 2+2 %.1f (x%.2f), 1+3 %.1f (x%.2f)\n",
           h2, h4, r11, h2 / r11, r22, h4 / r22, r13, h4 / r13, r22p, h4 / r22p, r13p, h4 / r13p);
    return 0;
}

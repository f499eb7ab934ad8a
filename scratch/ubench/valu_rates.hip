// Round 4: issue cost of the vector instructions the fused front end is made of, four waves per SIMD (1024 threads per CU),
// wall-clock per instruction and SIMD (hipEvents around 2000 trips of 64 unrolled instructions over 16 independent registers).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define BODY(ID, TXT)                                                                                                         \
    template <> __global__ void k<ID>(int iters, float *out) {                                                               \
        const int lane = threadIdx.x & 63;                                                                                    \
        float a[16];                                                                                                          \
        for (int i = 0; i < 16; ++i) a[i] = lane + i;                                                                         \
        float m = 1.0001f + lane, c = 0.5f + lane, d = 3.0f * lane;                                                           \
        float sm = __int_as_float(__builtin_amdgcn_readfirstlane(iters | 0x3f800000));                                        \
        asm volatile("" : "+v"(m), "+v"(c), "+v"(d));                                                                         \
        for (int it = 0; it < iters; ++it) {                                                                                  \
            _Pragma("unroll") for (int j = 0; j < 64; ++j) asm volatile(TXT : "+v"(a[j & 15]) : "v"(m), "v"(c), "v"(d), "s"(sm), "v"(a[(j + 5) & 15]), "v"(a[(j + 9) & 15])); \
        }                                                                                                                     \
        float r = 0;                                                                                                          \
        for (int i = 0; i < 16; ++i) r += a[i];                                                                               \
        out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                                       \
    }
template <int ID> __global__ void k(int iters, float *out);
BODY(0, "v_fma_f32 %0, %0, %1, %2")
BODY(1, "v_fma_f32 %0, %1, %2, %3")
BODY(2, "v_fma_f32 %0, %4, %0, %2")
BODY(3, "v_max_f32 %0, %0, %1")
BODY(4, "v_max_i32 %0, %0, %1")
BODY(5, "v_max3_f32 %0, %0, %1, %2")
BODY(6, "v_max3_i32 %0, %1, %2, %3")
BODY(7, "v_max3_i32 %0, %0, %5, %6")
BODY(8, "v_perm_b32 %0, %0, %1, %4")
BODY(9, "v_perm_b32 %0, %0, %1, %2")
BODY(10, "v_or_b32 %0, %0, %1")
BODY(11, "v_add_u32 %0, %0, %1")
BODY(12, "v_mov_b32 %0, %1")
BODY(13, "v_med3_f32 %0, %0, %1, %2")
BODY(14, "v_lshl_add_u32 %0, %0, 3, %1")
BODY(15, "v_add3_u32 %0, %0, %1, %2")
BODY(16, "v_mul_f32 %0, %0, %1")
BODY(17, "v_max_f32 %0, %5, %6")
BODY(18, "v_fma_f32 %0, %5, %6, %1")
BODY(19, "v_min3_f32 %0, %0, %5, %6")
BODY(20, "v_or3_b32 %0, %0, %5, %6")
BODY(21, "v_pk_max_i16 %0, %0, %1")
BODY(22, "v_cvt_f32_i32 %0, %1")
BODY(23, "v_mul_i32_i24 %0, %0, %1")
BODY(24, "v_mad_i32_i24 %0, %0, %1, %2")
BODY(25, "v_mul_lo_u32 %0, %0, %1")
BODY(26, "v_cndmask_b32 %0, %0, %1, vcc")
BODY(47, "v_cndmask_b32 %0, %0, %1, s[10:11]")
BODY(27, "v_bitop3_b32 %0, %0, %5, %6 bitop3:0x96")
BODY(28, "v_max_f32 %0, %0, %4")
BODY(29, "v_fma_f32 %0, %0, %4, %4")
BODY(30, "v_mov_b32_dpp %0, %1 row_shl:1 row_mask:0xf bank_mask:0xf")
BODY(31, "v_max_f32_sdwa %0, %5, %6 dst_sel:BYTE_1 dst_unused:UNUSED_PRESERVE src0_sel:DWORD src1_sel:DWORD")
BODY(32, "v_cvt_i32_f32_sdwa %0, %5 dst_sel:BYTE_2 dst_unused:UNUSED_PRESERVE src0_sel:DWORD")
BODY(33, "v_xor_b32 %0, %0, %1")
BODY(34, "v_lshlrev_b32 %0, 3, %0")
BODY(35, "v_and_b32 %0, %0, %1")
BODY(36, "v_sub_u32 %0, %0, %1")
BODY(37, "v_cmp_lt_f32 vcc, %0, %1")
BODY(38, "v_fma_f32 %0, %0, %1, 1.0")
BODY(39, "v_add_f32 %0, %0, %1")
BODY(40, "v_and_or_b32 %0, %0, %5, %6")
BODY(41, "v_lshl_or_b32 %0, %5, 8, %0")
BODY(42, "v_mul_f32 %0, %0, %4")
BODY(43, "v_add_u32 %0, %4, %0")
BODY(44, "v_ashrrev_i32 %0, 16, %0")
BODY(45, "v_max_f32 %0, |%5|, %0")
BODY(46, "v_cvt_i32_f32 %0, %1")

template <int ID>
static void run(const char *name) {
    const int iters = 2000, grid = 256;
    float *out;
    hipMalloc(&out, grid * 1024 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<ID>, dim3(grid), dim3(1024), 0, 0, iters, out);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    printf("%-50s %6.3f ns per instruction and SIMD (%5.2f cycles at 2.4 GHz)\n", name, best * 1e6 / iters / 256, best * 1e6 / iters / 256 * 2.4);
    hipFree(out);
}
int main() {
    run<0>("v_fma_f32 d, d, v, v");
    run<0>("v_fma_f32 d, d, v, v (again)");
    run<1>("v_fma_f32 d, v, v, v (three fixed sources)");
    run<2>("v_fma_f32 d, s, d, v");
    run<18>("v_fma_f32 d, x, y, v (rotating sources)");
    run<29>("v_fma_f32 d, d, s, s");
    run<16>("v_mul_f32 d, d, v");
    run<3>("v_max_f32 d, d, v");
    run<17>("v_max_f32 d, x, y");
    run<28>("v_max_f32 d, d, s");
    run<4>("v_max_i32 d, d, v");
    run<5>("v_max3_f32 d, d, v, v");
    run<19>("v_min3_f32 d, d, x, y");
    run<6>("v_max3_i32 d, v, v, v");
    run<7>("v_max3_i32 d, d, x, y");
    run<13>("v_med3_f32 d, d, v, v");
    run<8>("v_perm_b32 d, d, v, s");
    run<9>("v_perm_b32 d, d, v, v");
    run<10>("v_or_b32 d, d, v");
    run<20>("v_or3_b32 d, d, x, y");
    run<27>("v_bitop3_b32 d, d, x, y");
    run<11>("v_add_u32 d, d, v");
    run<15>("v_add3_u32 d, d, v, v");
    run<14>("v_lshl_add_u32 d, d, 3, v");
    run<12>("v_mov_b32 d, v");
    run<30>("v_mov_b32_dpp row_shl:1");
    run<21>("v_pk_max_i16 d, d, v");
    run<22>("v_cvt_f32_i32 d, v");
    run<23>("v_mul_i32_i24 d, d, v");
    run<24>("v_mad_i32_i24 d, d, v, v");
    run<25>("v_mul_lo_u32 d, d, v");
    run<26>("v_cndmask_b32 d, d, v, vcc");
    run<47>("v_cndmask_b32 d, d, v, s[10:11]");
    run<33>("v_xor_b32 d, d, v");
    run<34>("v_lshlrev_b32 d, 3, d");
    run<35>("v_and_b32 d, d, v");
    run<36>("v_sub_u32 d, d, v");
    run<37>("v_cmp_lt_f32 vcc, d, v");
    run<38>("v_fma_f32 d, d, v, 1.0");
    run<39>("v_add_f32 d, d, v");
    run<40>("v_and_or_b32 d, d, x, y");
    run<41>("v_lshl_or_b32 d, x, 8, d");
    run<42>("v_mul_f32 d, d, s");
    run<43>("v_add_u32 d, s, d");
    run<44>("v_ashrrev_i32 d, 16, d");
    run<45>("v_max_f32 d, |x|, d");
    run<46>("v_cvt_i32_f32 d, v");
    run<31>("v_max_f32_sdwa d.byte1, x, y (preserve)");
    run<32>("v_cvt_i32_f32_sdwa d.byte2, x (preserve)");
    return 0;
}

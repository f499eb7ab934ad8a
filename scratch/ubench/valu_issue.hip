// Round 4: what a SIMD issues per cycle when the front end's instruction mix runs on it.
// One workgroup of 64 * WPS * 4 threads per CU (WPS waves per SIMD), every wave runs `iters` trips of an unrolled body:
//   MODE 0  64 independent v_fma_f32                      MODE 1  64 independent v_pk_fma_f32 (128 fp32 fma)
//   MODE 2  64 v_max3_f32                                 MODE 3  64 v_perm_b32
//   MODE 4  64 v_max3_i32                                 MODE 5  8 MFMA 16x16x64 i8 alone
//   MODE 6  8 MFMA + 64 v_fma_f32 interleaved 1 : 8       MODE 7  8 MFMA + 64 v_fma_f32, MFMAs first
//   MODE 8  64 v_fma_f32 whose second source is an SGPR   MODE 9  32 v_fma_f32 + 32 s_add_u32 alternating
// Prints cycles (s_memtime) per trip and SIMD, median over the CUs' wave 0.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(int iters, float *out, unsigned long long *cyc, float sa) {
    const int lane = threadIdx.x & 63;
    float a[16];
    v2f p[16];
    v4i acc[8], wa = {lane, 1, 2, 3}, wb = {3, lane, 1, 0};
    unsigned int s0 = 1, s1 = 2;
    for (int i = 0; i < 16; ++i) { a[i] = lane + i; p[i] = (v2f){(float)lane, (float)i}; }
    for (int i = 0; i < 8; ++i) acc[i] = (v4i){0, 0, 0, 0};
    const float m = 1.0001f, c = 0.5f;
    const v2f m2 = {1.0001f, 0.9999f}, c2 = {0.5f, 0.25f};
    float ssrc = __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(sa)));
    __syncthreads();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if constexpr (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j & 15]) : "v"(m), "v"(c));
        } else if constexpr (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p[j & 15]) : "v"(m2), "v"(c2));
        } else if constexpr (MODE == 2) {
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(a[j & 15]) : "v"(m), "v"(c));
        } else if constexpr (MODE == 3) {
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(a[j & 15]) : "v"(m), "v"(c));
        } else if constexpr (MODE == 4) {
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_max3_i32 %0, %0, %1, %2" : "+v"(a[j & 15]) : "v"(m), "v"(c));
        } else if constexpr (MODE == 5) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(wa), "v"(wb));
        } else if constexpr (MODE == 6) {
#pragma unroll
            for (int j = 0; j < 64; ++j) {
                if ((j & 7) == 0) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[j >> 3]) : "v"(wa), "v"(wb));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j & 15]) : "v"(m), "v"(c));
            }
        } else if constexpr (MODE == 7) {
#pragma unroll
            for (int j = 0; j < 8; ++j) asm volatile("v_mfma_i32_16x16x64_i8 %0, %1, %2, %0" : "+v"(acc[j]) : "v"(wa), "v"(wb));
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j & 15]) : "v"(m), "v"(c));
        } else if constexpr (MODE == 8) {
#pragma unroll
            for (int j = 0; j < 64; ++j) asm volatile("v_fma_f32 %0, %1, %0, %2" : "+v"(a[j & 15]) : "s"(ssrc), "v"(c));
        } else {
#pragma unroll
            for (int j = 0; j < 32; ++j) {
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j & 15]) : "v"(m), "v"(c));
                asm volatile("s_add_u32 %0, %0, %1" : "+s"(s0) : "s"(s1) : "scc");
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float r = (float)s0;
    for (int i = 0; i < 16; ++i) r += a[i] + p[i][0] + p[i][1];
    for (int i = 0; i < 8; ++i) r += (float)(acc[i][0] + acc[i][3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
static void run(const char *name, int ninstr) {
    const int iters = 2000, grid = 256;
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, grid * 1024 * 4 * 2);
    hipMalloc(&cyc, grid * 8);
    printf("%-44s", name);
    for (int wps : {1, 2, 4}) {
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipMemset(cyc, 0, grid * 8);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256 * wps), 0, 0, iters, out, cyc, 1.0f);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<MODE>, dim3(grid), dim3(256 * wps), 0, 0, iters, out, cyc, 1.0f);
        hipEventRecord(e1, 0);
        hipError_t err = hipDeviceSynchronize();
        if (err != hipSuccess || hipGetLastError() != hipSuccess) printf(" [launch failed: %s]", hipGetErrorString(err));
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        std::vector<unsigned long long> h(grid);
        hipMemcpy(h.data(), cyc, grid * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        const double per_trip = (double)h[grid / 2] / iters;
        printf("  %d w/SIMD: %7.1f cyc/trip = %5.2f per instr/SIMD (%.1f ns/trip)", wps, per_trip, per_trip / (ninstr * wps), ms * 1e6 / iters);
    }
    printf("\n");
    hipFree(out);
    hipFree(cyc);
}

int main() {
    run<0>("64 v_fma_f32", 64);
    run<1>("64 v_pk_fma_f32", 64);
    run<2>("64 v_max3_f32", 64);
    run<3>("64 v_perm_b32", 64);
    run<4>("64 v_max3_i32", 64);
    run<8>("64 v_fma_f32 (SGPR source)", 64);
    run<9>("32 v_fma_f32 + 32 s_add_u32", 64);
    run<5>("8 MFMA 16x16x64 i8", 8);
    run<6>("8 MFMA + 64 v_fma interleaved (72)", 72);
    run<7>("8 MFMA then 64 v_fma (72)", 72);
    return 0;
}

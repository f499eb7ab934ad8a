// yolo355 -- y355_mfma_peak_i8: the int8 MFMA issue rate this GPU sustains, measured in the bench run itself.
// Not on the path: bench.py reports it beside the nominal 5.0 Pop/s its roofline fractions divide by (the
// "measured peak" used to be a constant).  Every wave keeps 24 independent accumulators and issues back-to-back
// v_mfma_i32_16x16x64_i8 on register operands holding pseudo-random bytes (zero operands let the clock rise:
// MI355X_MICROARCH.md, DVFS give-back), two waves per SIMD on every CU.
#include "../../include/yolo355.h"
#include "y355_common.h"
#include <string>

int y355_fail(int code, const std::string &msg);

namespace {
constexpr int NACC = 24;
__global__ __launch_bounds__(512) void mfma_peak_kernel(int iters, int *out, unsigned long long *cyc) {
    unsigned int hsh = (threadIdx.x + 1u) * 2654435761u ^ (blockIdx.x * 40503u);
    v4i a, b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        hsh = hsh * 1664525u + 1013904223u;
        a[i] = (int)hsh;
        hsh = hsh * 1664525u + 1013904223u;
        b[i] = (int)hsh;
    }
    v4i acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = (v4i){0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a, b, acc[i], 0, 0, 0);
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    int s = 0;
#pragma unroll
    for (int i = 0; i < NACC; ++i) s += acc[i][0] ^ acc[i][3];
    if (s == 0x7fffffff && iters < 0) out[0] = s;                 // keeps the accumulators alive
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}
}  // namespace

extern "C" int y355_mfma_peak_i8(int device_id, float ms_target, float *tops_out, float *clock_ghz_out) {
    if (!tops_out) return y355_fail(Y355_EINVAL, "null argument");
    if (hipSetDevice(device_id) != hipSuccess) return y355_fail(Y355_EHIP, "hipSetDevice failed");
    int *out = nullptr;
    unsigned long long *cyc = nullptr;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int rc = 0;
    auto chk = [&](hipError_t e, const char *what) {
        if (e != hipSuccess && !rc) rc = y355_fail(Y355_EHIP, std::string(what) + ": " + hipGetErrorString(e));
    };
    chk(hipMalloc((void **)&out, 4), "hipMalloc");
    chk(hipMalloc((void **)&cyc, 8), "hipMalloc");
    chk(hipEventCreate(&e0), "hipEventCreate");
    chk(hipEventCreate(&e1), "hipEventCreate");
    float ms = 0.f;
    int iters = 20000;
    if (!rc) {
        hipDeviceProp_t prop;
        chk(hipGetDeviceProperties(&prop, device_id), "hipGetDeviceProperties");
        const int wgs = prop.multiProcessorCount;
        // calibrate the trip count on a short run, then time `ms_target` of back-to-back issue
        for (int pass = 0; pass < 2 && !rc; ++pass) {
            chk(hipEventRecord(e0, 0), "hipEventRecord");
            hipLaunchKernelGGL(mfma_peak_kernel, dim3(wgs), dim3(512), 0, 0, iters, out, cyc);
            chk(hipEventRecord(e1, 0), "hipEventRecord");
            chk(hipEventSynchronize(e1), "hipEventSynchronize");
            chk(hipEventElapsedTime(&ms, e0, e1), "hipEventElapsedTime");
            if (pass == 0 && ms > 0.f) {
                double want = (double)iters * (ms_target > 0.f ? ms_target : 50.f) / ms;
                iters = (int)(want < 1000 ? 1000 : (want > 5e6 ? 5e6 : want));
            }
        }
        if (!rc) {
            unsigned long long c = 0;
            chk(hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost), "hipMemcpy");
            const double mfmas = (double)iters * NACC * 8.0 * wgs;
            *tops_out = (float)(mfmas * 2.0 * 16 * 16 * 64 / (ms * 1e-3) / 1e12);
            if (clock_ghz_out) *clock_ghz_out = (float)((double)c / (ms * 1e6));
        }
    }
    if (out) (void)hipFree(out);
    if (cyc) (void)hipFree(cyc);
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rc;
}

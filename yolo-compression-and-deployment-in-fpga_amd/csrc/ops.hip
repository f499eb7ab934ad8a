// yolo355 -- operator-level entry points for the two element-wise ops of the path that the fused
// layers absorb: stand-alone forms for unit tests and for callers that hold their own
// int8 tensors.  Host pointers, synchronous; NCHW like the reference's tensors.
//   y355_quantize_input_f32_i8  AveragedRangeTracker.quantize_activation on the network input
//                               (models/slim_yolo_v2.py:33-38): q = clamp(RNE(x * 2^sa), +-127)
//   y355_maxpool2x2_i8          nn.MaxPool2d(2, 2) (:61,65,71,77) on int8
#include "../../include/yolo355.h"
#include "y355_common.h"

#include <algorithm>
#include <cmath>
#include <string>

int y355_fail(int code, const std::string &msg);
#define OPSCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            (void)hipFree(d_in);                                                            \
            (void)hipFree(d_out);                                                           \
            (void)hipFree(d_cnt);                                                           \
            return y355_fail(Y355_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                   \
    } while (0)

namespace {
__global__ void quantize_f32_i8_kernel(const float *x, int8_t *q, size_t n, float scale, unsigned long long *nclamped) {
    unsigned int c = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float r = rintf(x[i] * scale);
        const float rc = fminf(fmaxf(r, -127.f), 127.f);
        c += rc != r ? 1u : 0u;
        q[i] = (int8_t)(int)rc;
    }
    if (c) atomicAdd(nclamped, (unsigned long long)c);
}

__global__ void maxpool2x2_i8_kernel(const int8_t *in, int8_t *out, size_t planes, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1;
    const size_t n = planes * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
        const size_t pl = i / ((size_t)Ho * Wo);
        const int8_t *s = in + (pl * H + 2 * y) * W + 2 * x;
        out[i] = (int8_t)max(max((int)s[0], (int)s[1]), max((int)s[W], (int)s[W + 1]));
    }
}
}  // namespace

extern "C" int y355_quantize_input_f32_i8(int device_id, const float *x, size_t n, int sa, int8_t *q, int64_t *clamped) {
    if (!x || !q) return y355_fail(Y355_EINVAL, "null argument");
    if (n == 0) return y355_fail(Y355_EINVAL, "empty tensor");
    if (sa < -64 || sa > 64) return y355_fail(Y355_EINVAL, "activation exponent out of range");
    float *d_in = nullptr;
    int8_t *d_out = nullptr;
    unsigned long long *d_cnt = nullptr;
    OPSCHK(hipSetDevice(device_id));
    OPSCHK(hipMalloc((void **)&d_in, n * sizeof(float)));
    OPSCHK(hipMalloc((void **)&d_out, n));
    OPSCHK(hipMalloc((void **)&d_cnt, 8));
    OPSCHK(hipMemcpy(d_in, x, n * sizeof(float), hipMemcpyHostToDevice));
    OPSCHK(hipMemset(d_cnt, 0, 8));
    const int blocks = (int)std::min<size_t>((n + 255) / 256, 8192);
    hipLaunchKernelGGL(quantize_f32_i8_kernel, dim3(blocks), dim3(256), 0, 0, d_in, d_out, n, std::ldexp(1.0f, sa), d_cnt);
    OPSCHK(hipGetLastError());
    OPSCHK(hipDeviceSynchronize());
    OPSCHK(hipMemcpy(q, d_out, n, hipMemcpyDeviceToHost));
    unsigned long long c = 0;
    OPSCHK(hipMemcpy(&c, d_cnt, 8, hipMemcpyDeviceToHost));
    if (clamped) *clamped = (int64_t)c;
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    (void)hipFree(d_cnt);
    return 0;
}

extern "C" int y355_maxpool2x2_i8(int device_id, const int8_t *in, int batch, int channels, int height, int width, int8_t *out) {
    if (!in || !out) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 1 || channels < 1 || height < 2 || width < 2) return y355_fail(Y355_EINVAL, "bad shape");
    if ((height | width) & 1) return y355_fail(Y355_EINVAL, "pooling needs even H, W");
    const size_t planes = (size_t)batch * channels, nin = planes * height * width, nout = nin / 4;
    int8_t *d_in = nullptr, *d_out = nullptr;
    unsigned long long *d_cnt = nullptr;
    OPSCHK(hipSetDevice(device_id));
    OPSCHK(hipMalloc((void **)&d_in, nin));
    OPSCHK(hipMalloc((void **)&d_out, nout));
    OPSCHK(hipMemcpy(d_in, in, nin, hipMemcpyHostToDevice));
    const int blocks = (int)std::min<size_t>((nout + 255) / 256, 8192);
    hipLaunchKernelGGL(maxpool2x2_i8_kernel, dim3(blocks), dim3(256), 0, 0, d_in, d_out, planes, height, width);
    OPSCHK(hipGetLastError());
    OPSCHK(hipDeviceSynchronize());
    OPSCHK(hipMemcpy(out, d_out, nout, hipMemcpyDeviceToHost));
    (void)hipFree(d_in);
    (void)hipFree(d_out);
    return 0;
}

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
#include <cstring>
#include <string>
#include <vector>

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

// ------------------------------------------------------------------------------------------
// Operator API of the wider model families, stand-alone forms.  Host pointers, fp32
// NCHW like the reference's tensors, synchronous.
//   y355_reorg_f32    utils.modules.reorg_layer (utils/modules.py:43-57): out[b][(sy*s+sx)*C + c][y][x] =
//                     in[b][c][s*y+sy][s*x+sx]  (data movement: bit-exact)
//   y355_spp_f32      utils.modules.SPP (:59-72): cat[x, maxpool5(x), maxpool9(x), maxpool13(x)], stride 1,
//                     padding k/2 with -inf  (max of fp32: bit-exact)
//   y355_conv2d_bf16  utils.modules.Conv2d / backbone.darknet.Conv_BN_LeakyReLU / resblock with BN folded
//                     by the caller: conv (1x1, or 3x3 pad 1; stride 1, or 2 for 3x3) + bias +
//                     LeakyReLU(neg_slope) [+ residual], operands rounded to bf16, fp32 accumulation on
//                     v_mfma_f32_16x16x32_bf16 (convg.hip), result rounded to bf16
namespace {
#define OPS2CHK(expr)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            for (void *q_ : bufs) (void)hipFree(q_);                                        \
            return y355_fail(Y355_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
        }                                                                                   \
    } while (0)

__global__ void reorg_f32_kernel(const float *in, float *out, int B, int C, int H, int W, int s) {
    const int Ho = H / s, Wo = W / s;
    const size_t n = (size_t)B * C * s * s * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
        const int oc = (int)((i / ((size_t)Wo * Ho)) % ((size_t)C * s * s));
        const int b = (int)(i / ((size_t)Wo * Ho * C * s * s));
        const int k = oc / C, c = oc % C, sy = k / s, sx = k % s;
        out[i] = in[(((size_t)b * C + c) * H + (size_t)s * y + sy) * W + (size_t)s * x + sx];
    }
}

__global__ void spp_f32_kernel(const float *in, float *out, size_t planes_b, int C, int H, int W) {
    // one thread per input element: writes x and the three window maxima (windows clipped to the map:
    // the -inf padding of max_pool2d never wins)
    const size_t n = planes_b * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((size_t)W * H)) % C);
        const size_t b = i / ((size_t)W * H * C);
        const float *pl = in + (b * C + c) * (size_t)H * W;
        float m5 = -INFINITY, m9 = -INFINITY, m13 = -INFINITY;
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -6; dx <= 6; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                const float v = pl[(size_t)yy * W + xx];
                const int r = max(abs(dy), abs(dx));
                m13 = fmaxf(m13, v);
                if (r <= 4) m9 = fmaxf(m9, v);
                if (r <= 2) m5 = fmaxf(m5, v);
            }
        }
        float *ob = out + b * 4 * C * (size_t)H * W + (size_t)y * W + x;
        const size_t cs = (size_t)H * W;
        ob[(size_t)c * cs] = pl[(size_t)y * W + x];
        ob[((size_t)C + c) * cs] = m5;
        ob[((size_t)2 * C + c) * cs] = m9;
        ob[((size_t)3 * C + c) * cs] = m13;
    }
}

// fp32 NCHW -> bf16 NHWC with a one-pixel halo and cpad channels (buffer zeroed beforehand)
__global__ void nchw_to_nhwc_bf16_kernel(const float *in, unsigned short *out, int B, int C, int H, int W, int cpad) {
    const size_t n = (size_t)B * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((size_t)W * H)) % C);
        const size_t b = i / ((size_t)W * H * C);
        out[((b * (H + 2) + y + 1) * (size_t)(W + 2) + x + 1) * cpad + c] = __builtin_bit_cast(unsigned short, (__bf16)in[i]);
    }
}
__global__ void nhwc_bf16_to_nchw_kernel(const unsigned short *in, float *out, int B, int C, int H, int W, int cpad) {
    const size_t n = (size_t)B * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((size_t)W * H)) % C);
        const size_t b = i / ((size_t)W * H * C);
        out[i] = __uint_as_float((unsigned int)in[((b * (H + 2) + y + 1) * (size_t)(W + 2) + x + 1) * cpad + c] << 16);
    }
}
__global__ void nhwc_f32_to_nchw_kernel(const float *in, float *out, int B, int C, int H, int W, int cpad) {
    const size_t n = (size_t)B * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((size_t)W * H)) % C);
        const size_t b = i / ((size_t)W * H * C);
        out[i] = in[((b * (H + 2) + y + 1) * (size_t)(W + 2) + x + 1) * cpad + c];
    }
}
__global__ void maxpool2x2_f32_kernel(const float *in, float *out, size_t planes, int H, int W) {
    const int Ho = H >> 1, Wo = W >> 1;
    const size_t n = planes * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
        const size_t pl = i / ((size_t)Ho * Wo);
        const float *s = in + (pl * H + 2 * y) * W + 2 * x;
        out[i] = fmaxf(fmaxf(s[0], s[1]), fmaxf(s[W], s[W + 1]));
    }
}
// F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True) (models/yolo_v3.py:211,215)
__global__ void upsample2x_f32_kernel(const float *in, float *out, size_t planes, int H, int W) {
    const int Ho = 2 * H, Wo = 2 * W;
    const float ry = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f, rx = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const size_t n = planes * Ho * Wo;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % Wo), y = (int)((i / Wo) % Ho);
        const size_t pl = i / ((size_t)Ho * Wo);
        const float sy = ry * (float)y, sx = rx * (float)x;
        const int y0 = min((int)sy, H - 1), x0 = min((int)sx, W - 1);
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float *s = in + pl * (size_t)H * W;
        const float top = (1.f - lx) * s[(size_t)y0 * W + x0] + lx * s[(size_t)y0 * W + x1];
        const float bot = (1.f - lx) * s[(size_t)y1 * W + x0] + lx * s[(size_t)y1 * W + x1];
        out[i] = (1.f - ly) * top + ly * bot;
    }
}
inline int grid_for(size_t n) { return (int)std::min<size_t>((n + 255) / 256, 16384); }
}  // namespace

extern "C" int y355_reorg_f32(int device_id, const float *x, int batch, int channels, int height, int width, int stride, float *out) {
    if (!x || !out) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 1 || channels < 1 || stride < 1 || height < stride || width < stride) return y355_fail(Y355_EINVAL, "bad shape");
    if (height % stride || width % stride) return y355_fail(Y355_EINVAL, "reorg needs H, W divisible by the stride");
    const size_t n = (size_t)batch * channels * height * width;
    std::vector<void *> bufs;
    float *d_in = nullptr, *d_out = nullptr;
    OPS2CHK(hipSetDevice(device_id));
    OPS2CHK(hipMalloc((void **)&d_in, n * 4)); bufs.push_back(d_in);
    OPS2CHK(hipMalloc((void **)&d_out, n * 4)); bufs.push_back(d_out);
    OPS2CHK(hipMemcpy(d_in, x, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(reorg_f32_kernel, dim3(grid_for(n)), dim3(256), 0, 0, d_in, d_out, batch, channels, height, width, stride);
    OPS2CHK(hipGetLastError());
    OPS2CHK(hipDeviceSynchronize());
    OPS2CHK(hipMemcpy(out, d_out, n * 4, hipMemcpyDeviceToHost));
    for (void *q : bufs) (void)hipFree(q);
    return 0;
}

extern "C" int y355_spp_f32(int device_id, const float *x, int batch, int channels, int height, int width, float *out) {
    if (!x || !out) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 1 || channels < 1 || height < 1 || width < 1) return y355_fail(Y355_EINVAL, "bad shape");
    const size_t n = (size_t)batch * channels * height * width;
    std::vector<void *> bufs;
    float *d_in = nullptr, *d_out = nullptr;
    OPS2CHK(hipSetDevice(device_id));
    OPS2CHK(hipMalloc((void **)&d_in, n * 4)); bufs.push_back(d_in);
    OPS2CHK(hipMalloc((void **)&d_out, n * 16)); bufs.push_back(d_out);
    OPS2CHK(hipMemcpy(d_in, x, n * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(spp_f32_kernel, dim3(grid_for(n)), dim3(256), 0, 0, d_in, d_out, (size_t)batch, channels, height, width);
    OPS2CHK(hipGetLastError());
    OPS2CHK(hipDeviceSynchronize());
    OPS2CHK(hipMemcpy(out, d_out, n * 16, hipMemcpyDeviceToHost));
    for (void *q : bufs) (void)hipFree(q);
    return 0;
}

extern "C" int y355_conv2d_bf16(int device_id, const float *x, const float *w, const float *bias, const float *residual,
                                int batch, int cin, int cout, int height, int width, int ksize, int stride, float neg_slope,
                                int out_fp32, float *out) {
    if (!x || !w || !out) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 1 || cin < 1 || cout < 1 || height < 1 || width < 1) return y355_fail(Y355_EINVAL, "bad shape");
    if (ksize != 1 && ksize != 3) return y355_fail(Y355_EINVAL, "kernel size 1 or 3 (padding k/2)");
    if (stride != 1 && !(stride == 2 && ksize == 3)) return y355_fail(Y355_EINVAL, "stride 1, or 2 with a 3x3 kernel");
    const bool thin = (cin <= 16 && ksize == 3 && stride == 1);
    const int cin_pad = thin ? 16 : (cin + 31) / 32 * 32;
    const int in_pb = cin_pad * 2;
    const int kid = y355_convg_select(in_pb, cout, 0, height, width, stride);
    const ConvGInfo *ki = y355_convg_kernel(1, kid);
    if (!ki) return y355_fail(Y355_EINVAL, "no kernel for this shape");
    const int Ho = stride == 2 ? (height + 1) / 2 : height, Wo = stride == 2 ? (width + 1) / 2 : width;
    const int cout_pad = (cout + ki->bn - 1) / ki->bn * ki->bn;
    const int taps = ksize * ksize;
    const size_t wbytes = y355_convg_packed_bytes(*ki, in_pb, taps, cout_pad);
    std::vector<char> wpk(wbytes);
    y355_convg_pack(*ki, w, nullptr, cout, cin, ksize, in_pb, cout_pad, wpk.data());
    std::vector<float> bpad(cout_pad, 0.f);
    if (bias) std::copy(bias, bias + cout, bpad.begin());
    const size_t n_in = (size_t)batch * cin * height * width, n_out = (size_t)batch * cout * Ho * Wo;
    const size_t in_bytes = (size_t)batch * (height + 2) * (width + 2) * in_pb;
    if (out_fp32 && residual) return y355_fail(Y355_EINVAL, "fp32 output (prediction layers) takes no residual");
    const size_t out_pb = (size_t)cout_pad * (out_fp32 ? 4 : 2), out_bytes = (size_t)batch * (Ho + 2) * (Wo + 2) * out_pb;
    std::vector<void *> bufs;
    float *d_x = nullptr, *d_y = nullptr, *d_b = nullptr;
    char *d_in = nullptr, *d_out = nullptr, *d_w = nullptr, *d_res = nullptr;
    OPS2CHK(hipSetDevice(device_id));
    if (int e = y355_prepare_convg()) return y355_fail(Y355_EHIP, std::string("kernel attributes: ") + hipGetErrorString((hipError_t)e));
    OPS2CHK(hipMalloc((void **)&d_x, std::max(n_in, n_out) * 4)); bufs.push_back(d_x);
    OPS2CHK(hipMalloc((void **)&d_y, n_out * 4)); bufs.push_back(d_y);
    OPS2CHK(hipMalloc((void **)&d_b, cout_pad * 4)); bufs.push_back(d_b);
    OPS2CHK(hipMalloc((void **)&d_in, in_bytes)); bufs.push_back(d_in);
    OPS2CHK(hipMalloc((void **)&d_out, out_bytes)); bufs.push_back(d_out);
    OPS2CHK(hipMalloc((void **)&d_w, wbytes)); bufs.push_back(d_w);
    OPS2CHK(hipMemset(d_in, 0, in_bytes));
    OPS2CHK(hipMemset(d_out, 0, out_bytes));
    OPS2CHK(hipMemcpy(d_w, wpk.data(), wbytes, hipMemcpyHostToDevice));
    OPS2CHK(hipMemcpy(d_b, bpad.data(), cout_pad * 4, hipMemcpyHostToDevice));
    if (residual) {
        OPS2CHK(hipMalloc((void **)&d_res, out_bytes)); bufs.push_back(d_res);
        OPS2CHK(hipMemset(d_res, 0, out_bytes));
        OPS2CHK(hipMemcpy(d_x, residual, n_out * 4, hipMemcpyHostToDevice));
        hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3(grid_for(n_out)), dim3(256), 0, 0, d_x, (unsigned short *)d_res, batch, cout, Ho,
                           Wo, cout_pad);
        OPS2CHK(hipDeviceSynchronize());
    }
    OPS2CHK(hipMemcpy(d_x, x, n_in * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3(grid_for(n_in)), dim3(256), 0, 0, d_x, (unsigned short *)d_in, batch, cin, height,
                       width, cin_pad);
    ConvGParams p{};
    p.in = d_in;
    p.out = d_out;
    p.w = d_w;
    p.bias_f = d_b;
    p.B = batch;
    p.H = height;
    p.W = width;
    p.in_pb = in_pb;
    p.nchunks = in_pb / ki->chb;
    p.out_pb = (int)out_pb;
    p.out_off = 0;
    p.out_halo = 1;
    p.tiles_x = (Wo + ki->tw - 1) / ki->tw;
    p.tiles_y = (Ho + ki->th - 1) / ki->th;
    p.nblk = cout_pad / ki->bn;
    p.taps = taps;
    p.slope = neg_slope;
    p.out_f32 = out_fp32 ? 1 : 0;
    p.res = d_res;
    p.res_pb = (int)out_pb;
    p.res_off = 0;
    ki->launch(p, p.tiles_x * p.tiles_y * p.nblk * batch, 0);
    OPS2CHK(hipGetLastError());
    if (out_fp32)
        hipLaunchKernelGGL(nhwc_f32_to_nchw_kernel, dim3(grid_for(n_out)), dim3(256), 0, 0, (const float *)d_out, d_y, batch, cout, Ho, Wo,
                           cout_pad);
    else
        hipLaunchKernelGGL(nhwc_bf16_to_nchw_kernel, dim3(grid_for(n_out)), dim3(256), 0, 0, (const unsigned short *)d_out, d_y, batch,
                           cout, Ho, Wo, cout_pad);
    OPS2CHK(hipGetLastError());
    OPS2CHK(hipDeviceSynchronize());
    OPS2CHK(hipMemcpy(out, d_y, n_out * 4, hipMemcpyDeviceToHost));
    for (void *q : bufs) (void)hipFree(q);
    return 0;
}

extern "C" int y355_maxpool2x2_f32(int device_id, const float *in, int batch, int channels, int height, int width, float *out) {
    if (!in || !out) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 1 || channels < 1 || height < 2 || width < 2) return y355_fail(Y355_EINVAL, "bad shape");
    if ((height | width) & 1) return y355_fail(Y355_EINVAL, "pooling needs even H, W");
    const size_t planes = (size_t)batch * channels, nin = planes * height * width, nout = nin / 4;
    std::vector<void *> bufs;
    float *d_in = nullptr, *d_out = nullptr;
    OPS2CHK(hipSetDevice(device_id));
    OPS2CHK(hipMalloc((void **)&d_in, nin * 4)); bufs.push_back(d_in);
    OPS2CHK(hipMalloc((void **)&d_out, nout * 4)); bufs.push_back(d_out);
    OPS2CHK(hipMemcpy(d_in, in, nin * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(maxpool2x2_f32_kernel, dim3(grid_for(nout)), dim3(256), 0, 0, d_in, d_out, planes, height, width);
    OPS2CHK(hipGetLastError());
    OPS2CHK(hipDeviceSynchronize());
    OPS2CHK(hipMemcpy(out, d_out, nout * 4, hipMemcpyDeviceToHost));
    for (void *q : bufs) (void)hipFree(q);
    return 0;
}

extern "C" int y355_upsample2x_f32(int device_id, const float *in, int batch, int channels, int height, int width, float *out) {
    if (!in || !out) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 1 || channels < 1 || height < 1 || width < 1) return y355_fail(Y355_EINVAL, "bad shape");
    const size_t planes = (size_t)batch * channels, nin = planes * height * width, nout = nin * 4;
    std::vector<void *> bufs;
    float *d_in = nullptr, *d_out = nullptr;
    OPS2CHK(hipSetDevice(device_id));
    OPS2CHK(hipMalloc((void **)&d_in, nin * 4)); bufs.push_back(d_in);
    OPS2CHK(hipMalloc((void **)&d_out, nout * 4)); bufs.push_back(d_out);
    OPS2CHK(hipMemcpy(d_in, in, nin * 4, hipMemcpyHostToDevice));
    hipLaunchKernelGGL(upsample2x_f32_kernel, dim3(grid_for(nout)), dim3(256), 0, 0, d_in, d_out, planes, height, width);
    OPS2CHK(hipGetLastError());
    OPS2CHK(hipDeviceSynchronize());
    OPS2CHK(hipMemcpy(out, d_out, nout * 4, hipMemcpyDeviceToHost));
    for (void *q : bufs) (void)hipFree(q);
    return 0;
}


// ==========================================================================================================================
// Device-resident operator forms (round 6).  The entry points above take host pointers and copy through
// temporary device buffers -- right for unit tests, wrong for a caller whose tensors already live on the GPU
// (utils.modules.Conv2d / Conv2d_fuse / reorg_layer / SPP called on CUDA tensors).  These take DEVICE pointers (fp32 NCHW, the
// reference's tensors) and a stream, launch behind whatever that stream holds and return without a host round trip.
//   y355_reorg_f32_dev / y355_spp_f32_dev / y355_maxpool2x2_f32_dev / y355_upsample2x_f32_dev: one launch each
//   y355_conv_op: a convolution whose weights were packed onto the device ONCE (create), then forward per call;
//     bf16 form = y355_conv2d_bf16's arithmetic; int8 form = y355_conv3x3_i8_raw's (Conv2d_fuse on dyadic operands, exact):
//     the input's exponent and the verdict "is x a dyadic int8 tensor" are needed on the HOST (they select the route and the
//     requantisation constants): two 4-byte read-backs per call, nothing else leaves the device.
// A y355_conv_op owns its packed weights and growable workspaces; single-threaded like an engine handle.
namespace {
// fp32 NCHW dyadic tensor -> int8 NHWC with halo (cpad channels): q = x * 2^sa; *bad += values that are not integers in [-127, 127]
__global__ void dyadic_to_nhwc_i8_kernel(const float *in, int8_t *out, int B, int C, int H, int W, int cpad, float scale, unsigned int *bad) {
    const size_t n = (size_t)B * C * H * W;
    unsigned int nb = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((size_t)W * H)) % C);
        const size_t b = i / ((size_t)W * H * C);
        const float v = in[i] * scale, r = rintf(v);
        nb += (r != v || !(fabsf(r) <= 127.f)) ? 1u : 0u;
        out[((b * (H + 2) + y + 1) * (size_t)(W + 2) + x + 1) * cpad + c] = (int8_t)(int)fminf(fmaxf(r, -127.f), 127.f);
    }
    if (nb) atomicAdd(bad, nb);
}
// t' [B][H][W][cpad] int64 -> fp32 NCHW: t' * 2^-F' (exact wherever the reference's own fp32 result is)
__global__ void raw_to_nchw_f32_kernel(const long long *raw, float *out, int B, int C, int H, int W, int cpad, float inv) {
    const size_t n = (size_t)B * C * H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int x = (int)(i % W), y = (int)((i / W) % H);
        const int c = (int)((i / ((size_t)W * H)) % C);
        const size_t b = i / ((size_t)W * H * C);
        out[i] = (float)raw[((b * H + y) * (size_t)W + x) * cpad + c] * inv;
    }
}
#define DEVCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) return y355_fail(Y355_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)
}  // namespace
int y355_prepare_kernels();
int y355_op_requant(int cin, int sa_in, int e_w, int e_b, int act, const int32_t *q_b, int cout, int cout_pad, Requant *rq,
                    int *frac_bits, std::vector<int32_t> *bias_t, std::vector<long long> *bias_w);       // engine.hip

extern "C" int y355_reorg_f32_dev(const float *x_dev, int batch, int channels, int height, int width, int stride, float *out_dev, void *stream) {
    if (!x_dev || !out_dev) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 1 || channels < 1 || stride < 1 || height < stride || width < stride) return y355_fail(Y355_EINVAL, "bad shape");
    if (height % stride || width % stride) return y355_fail(Y355_EINVAL, "reorg needs H, W divisible by the stride");
    const size_t n = (size_t)batch * channels * height * width;
    hipLaunchKernelGGL(reorg_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x_dev, out_dev, batch, channels, height, width, stride);
    DEVCHK(hipGetLastError());
    return 0;
}
extern "C" int y355_spp_f32_dev(const float *x_dev, int batch, int channels, int height, int width, float *out_dev, void *stream) {
    if (!x_dev || !out_dev) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 1 || channels < 1 || height < 1 || width < 1) return y355_fail(Y355_EINVAL, "bad shape");
    const size_t n = (size_t)batch * channels * height * width;
    hipLaunchKernelGGL(spp_f32_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, x_dev, out_dev, (size_t)batch, channels, height, width);
    DEVCHK(hipGetLastError());
    return 0;
}
extern "C" int y355_maxpool2x2_f32_dev(const float *in_dev, int batch, int channels, int height, int width, float *out_dev, void *stream) {
    if (!in_dev || !out_dev) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 1 || channels < 1 || height < 2 || width < 2 || ((height | width) & 1)) return y355_fail(Y355_EINVAL, "bad shape (even H, W)");
    const size_t planes = (size_t)batch * channels, nout = planes * height * width / 4;
    hipLaunchKernelGGL(maxpool2x2_f32_kernel, dim3(grid_for(nout)), dim3(256), 0, (hipStream_t)stream, in_dev, out_dev, planes, height, width);
    DEVCHK(hipGetLastError());
    return 0;
}
extern "C" int y355_upsample2x_f32_dev(const float *in_dev, int batch, int channels, int height, int width, float *out_dev, void *stream) {
    if (!in_dev || !out_dev) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 1 || channels < 1 || height < 1 || width < 1) return y355_fail(Y355_EINVAL, "bad shape");
    const size_t planes = (size_t)batch * channels, nout = planes * height * width * 4;
    hipLaunchKernelGGL(upsample2x_f32_kernel, dim3(grid_for(nout)), dim3(256), 0, (hipStream_t)stream, in_dev, out_dev, planes, height, width);
    DEVCHK(hipGetLastError());
    return 0;
}

struct y355_conv_op {
    int device = 0, kind = 0;          // kind 0 bf16, 1 int8
    int cin = 0, cout = 0, ksize = 3, stride = 1;
    float slope = 1.f;
    // bf16
    int cin_pad = 0;
    char *w_dev = nullptr;             // y355_convg_pack layouts, one per kernel id the shapes select (packed on demand from w_host)
    int w_kid = -1, w_cout_pad = 0;
    std::vector<float> w_host, b_host;
    float *bias_dev = nullptr;
    // int8
    std::vector<int8_t> qw;
    std::vector<int32_t> qb;
    int e_w = 0, e_b = 0, act = 0, cpad = 0, cout_pad8 = 0;
    int8_t *qw_dev = nullptr;
    int *bt_dev = nullptr;
    long long *bw_dev = nullptr;
    int sa_cached = 1 << 30;
    Requant rq{};
    int frac_bits = 0;
    Counters *ctr_dev = nullptr;
    unsigned int *flag_dev = nullptr;  // [0] absmax bits, [1] non-dyadic count
    // workspaces (grown on demand)
    char *in_dev = nullptr, *out_dev = nullptr, *res_dev = nullptr;
    size_t in_cap = 0, out_cap = 0, res_cap = 0;
    size_t geo = 0;                    // (batch, H, W, output type) the halo buffers were last zeroed for
};

static int op_grow(char **p, size_t *cap, size_t need, bool zero) {
    if (*cap >= need) return 0;
    if (*p) (void)hipFree(*p);
    *p = nullptr;
    *cap = 0;
    DEVCHK(hipMalloc((void **)p, need));
    if (zero) DEVCHK(hipMemset(*p, 0, need));
    *cap = need;
    return 0;
}

extern "C" void y355_conv_op_destroy(y355_conv_op *op) {
    if (!op) return;
    (void)hipSetDevice(op->device);
    (void)hipDeviceSynchronize();
    for (void *q : {(void *)op->w_dev, (void *)op->bias_dev, (void *)op->qw_dev, (void *)op->bt_dev, (void *)op->bw_dev, (void *)op->ctr_dev,
                    (void *)op->flag_dev, (void *)op->in_dev, (void *)op->out_dev, (void *)op->res_dev})
        if (q) (void)hipFree(q);
    delete op;
}

extern "C" int y355_conv_op_create_bf16(int device_id, const float *w, const float *bias, int cin, int cout, int ksize, int stride,
                                        float neg_slope, y355_conv_op **out) {
    if (!w || !out) return y355_fail(Y355_EINVAL, "null argument");
    if (cin < 1 || cout < 1) return y355_fail(Y355_EINVAL, "bad shape");
    if (ksize != 1 && ksize != 3) return y355_fail(Y355_EINVAL, "kernel size 1 or 3 (padding k/2)");
    if (stride != 1 && !(stride == 2 && ksize == 3)) return y355_fail(Y355_EINVAL, "stride 1, or 2 with a 3x3 kernel");
    DEVCHK(hipSetDevice(device_id));
    if (int e = y355_prepare_convg()) return y355_fail(Y355_EHIP, std::string("kernel attributes: ") + hipGetErrorString((hipError_t)e));
    y355_conv_op *op = new y355_conv_op();
    op->device = device_id;
    op->kind = 0;
    op->cin = cin; op->cout = cout; op->ksize = ksize; op->stride = stride; op->slope = neg_slope;
    const bool thin = (cin <= 16 && ksize == 3 && stride == 1);
    op->cin_pad = thin ? 16 : (cin + 31) / 32 * 32;
    op->w_host.assign(w, w + (size_t)cout * cin * ksize * ksize);
    op->b_host.assign(cout, 0.f);
    if (bias) std::copy(bias, bias + cout, op->b_host.begin());
    *out = op;
    return 0;
}

extern "C" int y355_conv_op_forward(y355_conv_op *op, const float *x_dev, const float *residual_dev, int batch, int height, int width,
                                    int out_fp32, float *out_dev, void *stream_) {
    if (!op || op->kind != 0 || !x_dev || !out_dev) return y355_fail(Y355_EINVAL, "null argument / not a bf16 operator");
    if (batch < 1 || height < 1 || width < 1) return y355_fail(Y355_EINVAL, "bad shape");
    if (out_fp32 && residual_dev) return y355_fail(Y355_EINVAL, "fp32 output (prediction layers) takes no residual");
    hipStream_t s = (hipStream_t)stream_;
    DEVCHK(hipSetDevice(op->device));
    const int in_pb = op->cin_pad * 2, taps = op->ksize * op->ksize, stride = op->stride;
    const int kid = y355_convg_select(in_pb, op->cout, 0, height, width, stride);
    const ConvGInfo *ki = y355_convg_kernel(1, kid);
    if (!ki) return y355_fail(Y355_EINVAL, "no kernel for this shape");
    const int Ho = stride == 2 ? (height + 1) / 2 : height, Wo = stride == 2 ? (width + 1) / 2 : width;
    const int cout_pad = (op->cout + ki->bn - 1) / ki->bn * ki->bn;
    if (op->w_kid != kid || op->w_cout_pad != cout_pad) {          // first call, or a map size that selects another tile shape: (re)pack
        const size_t wbytes = y355_convg_packed_bytes(*ki, in_pb, taps, cout_pad);
        std::vector<char> wpk(wbytes);
        y355_convg_pack(*ki, op->w_host.data(), nullptr, op->cout, op->cin, op->ksize, in_pb, cout_pad, wpk.data());
        std::vector<float> bpad(cout_pad, 0.f);
        std::copy(op->b_host.begin(), op->b_host.end(), bpad.begin());
        DEVCHK(hipStreamSynchronize(s));                           // a previous forward may still read the old fragments
        if (op->w_dev) (void)hipFree(op->w_dev);
        if (op->bias_dev) (void)hipFree(op->bias_dev);
        op->w_dev = nullptr; op->bias_dev = nullptr;
        DEVCHK(hipMalloc((void **)&op->w_dev, wbytes));
        DEVCHK(hipMalloc((void **)&op->bias_dev, sizeof(float) * cout_pad));
        DEVCHK(hipMemcpy(op->w_dev, wpk.data(), wbytes, hipMemcpyHostToDevice));
        DEVCHK(hipMemcpy(op->bias_dev, bpad.data(), sizeof(float) * cout_pad, hipMemcpyHostToDevice));
        op->w_kid = kid;
        op->w_cout_pad = cout_pad;
        op->geo = 0;                                               // halo layouts depend on the padded channel counts: re-zero below
    }
    const size_t n_in = (size_t)batch * op->cin * height * width, n_out = (size_t)batch * op->cout * Ho * Wo;
    const size_t in_bytes = (size_t)batch * (height + 2) * (width + 2) * in_pb;
    const size_t out_pb = (size_t)cout_pad * (out_fp32 ? 4 : 2), out_bytes = (size_t)batch * (Ho + 2) * (Wo + 2) * out_pb;
    // a buffer is zeroed when it is (re)allocated: the staging kernels only write the interior, so the halo and the padding
    // channels stay zero for every later call of the same or a smaller size... as long as the geometry is the same: a new
    // geometry re-zeroes (cheap next to a reallocation)
    const size_t geo = ((size_t)batch << 40) ^ ((size_t)height << 20) ^ (size_t)width ^ ((size_t)out_fp32 << 62) ^ ((size_t)1 << 61);
    const bool regeo = geo != op->geo;
    op->geo = geo;
    if (int rc = op_grow(&op->in_dev, &op->in_cap, in_bytes, true)) return rc;
    if (int rc = op_grow(&op->out_dev, &op->out_cap, out_bytes, true)) return rc;
    if (regeo) {
        DEVCHK(hipMemsetAsync(op->in_dev, 0, in_bytes, s));
        DEVCHK(hipMemsetAsync(op->out_dev, 0, out_bytes, s));
    }
    if (residual_dev) {
        if (int rc = op_grow(&op->res_dev, &op->res_cap, out_bytes, true)) return rc;
        if (regeo) DEVCHK(hipMemsetAsync(op->res_dev, 0, out_bytes, s));
        hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3(grid_for(n_out)), dim3(256), 0, s, residual_dev, (unsigned short *)op->res_dev, batch,
                           op->cout, Ho, Wo, cout_pad);
    }
    hipLaunchKernelGGL(nchw_to_nhwc_bf16_kernel, dim3(grid_for(n_in)), dim3(256), 0, s, x_dev, (unsigned short *)op->in_dev, batch, op->cin, height,
                       width, op->cin_pad);
    ConvGParams p{};
    p.in = op->in_dev;
    p.out = op->out_dev;
    p.w = op->w_dev;
    p.bias_f = op->bias_dev;
    p.B = batch;
    p.H = height;
    p.W = width;
    p.in_pb = in_pb;
    p.nchunks = in_pb / ki->chb;
    p.out_pb = (int)out_pb;
    p.out_off = 0;
    p.out_halo = 1;
    p.tiles_x = (Wo + ki->tw - 1) / ki->tw;
    p.tiles_y = (Ho + ki->th - 1) / ki->th;
    p.nblk = cout_pad / ki->bn;
    p.taps = taps;
    p.slope = op->slope;
    p.out_f32 = out_fp32 ? 1 : 0;
    p.res = residual_dev ? op->res_dev : nullptr;
    p.res_pb = (int)out_pb;
    p.res_off = 0;
    ki->launch(p, p.tiles_x * p.tiles_y * p.nblk * batch, s);
    DEVCHK(hipGetLastError());
    if (out_fp32)
        hipLaunchKernelGGL(nhwc_f32_to_nchw_kernel, dim3(grid_for(n_out)), dim3(256), 0, s, (const float *)op->out_dev, out_dev, batch, op->cout, Ho,
                           Wo, cout_pad);
    else
        hipLaunchKernelGGL(nhwc_bf16_to_nchw_kernel, dim3(grid_for(n_out)), dim3(256), 0, s, (const unsigned short *)op->out_dev, out_dev, batch,
                           op->cout, Ho, Wo, cout_pad);
    DEVCHK(hipGetLastError());
    return 0;
}

extern "C" int y355_conv_op_create_i8(int device_id, const int8_t *q_w, const int32_t *q_b, int cin, int cout, int e_w, int e_b, int flags,
                                      y355_conv_op **out) {
    if (!q_w || !q_b || !out) return y355_fail(Y355_EINVAL, "null argument");
    if (cin < 1 || cin > 256 || cout < 1) return y355_fail(Y355_EINVAL, "bad shape (cin <= 256)");
    if ((flags & Y355_OP_LEAKY) && (flags & Y355_OP_RELU)) return y355_fail(Y355_EINVAL, "LeakyReLU and ReLU are exclusive");
    DEVCHK(hipSetDevice(device_id));
    if (int e = y355_prepare_kernels()) return e;
    y355_conv_op *op = new y355_conv_op();
    op->device = device_id;
    op->kind = 1;
    op->cin = cin; op->cout = cout;
    op->e_w = e_w; op->e_b = e_b;
    op->act = (flags & Y355_OP_LEAKY) ? 1 : ((flags & Y355_OP_RELU) ? 2 : 0);
    op->cpad = cin <= 16 ? 16 : cin <= 32 ? 32 : cin <= 64 ? 64 : cin <= 128 ? 128 : 256;
    const int sel = op->cpad == 16 ? 0 : op->cpad == 32 ? 1 : op->cpad == 64 ? 2 : op->cpad == 128 ? 3 : 4;
    const ConvKernelInfo &ki = *y355_conv_kernel(Y355_K_GEN16 + sel);
    op->cout_pad8 = (cout + ki.bn - 1) / ki.bn * ki.bn;
    op->qw.assign(q_w, q_w + (size_t)cout * cin * 9);
    op->qb.assign(q_b, q_b + cout);
    std::vector<int8_t> packed(y355_packed_bytes(ki, op->cout_pad8));
    y355_pack_weights(ki, q_w, cout, cin, op->cout_pad8, packed.data());
    auto bail = [&](int rc) { std::string keep = y355_last_error(); y355_conv_op_destroy(op); return y355_fail(rc, keep); };
    if (hipMalloc((void **)&op->qw_dev, packed.size()) != hipSuccess || hipMalloc((void **)&op->bt_dev, sizeof(int) * op->cout_pad8) != hipSuccess ||
        hipMalloc((void **)&op->bw_dev, sizeof(long long) * op->cout_pad8) != hipSuccess || hipMalloc((void **)&op->ctr_dev, sizeof(Counters)) != hipSuccess ||
        hipMalloc((void **)&op->flag_dev, 16) != hipSuccess ||
        hipMemcpy(op->qw_dev, packed.data(), packed.size(), hipMemcpyHostToDevice) != hipSuccess) {
        y355_fail(Y355_EHIP, "device allocation failed");
        return bail(Y355_EHIP);
    }
    *out = op;
    return 0;
}

// Conv2d_fuse(x) for a dyadic x (utils/modules.py:20-29 on the fake-quantised operands of the quantized path): exact.
// *exact = 0: x is not a dyadic int8 tensor (out_dev untouched) -- the caller takes the bf16 route.
extern "C" int y355_conv_op_forward_i8(y355_conv_op *op, const float *x_dev, int batch, int height, int width, float *out_dev, void *stream_,
                                       int32_t *sa_in, int32_t *exact) {
    if (!op || op->kind != 1 || !x_dev || !out_dev || !exact) return y355_fail(Y355_EINVAL, "null argument / not an int8 operator");
    if (batch < 1 || height < 1 || width < 1) return y355_fail(Y355_EINVAL, "bad shape");
    hipStream_t s = (hipStream_t)stream_;
    DEVCHK(hipSetDevice(op->device));
    *exact = 0;
    const size_t n_in = (size_t)batch * op->cin * height * width, n_out = (size_t)batch * op->cout * height * width;
    // (1) the exponent of x: floor(log2(127 / max|x|)) -- the tensor is q / 2^e with |q| <= 127 (prep.as_dyadic_int8)
    DEVCHK(hipMemsetAsync(op->flag_dev, 0, 16, s));
    y355_launch_absmax(x_dev, n_in, op->flag_dev, s);
    DEVCHK(hipGetLastError());
    unsigned int bits = 0;
    DEVCHK(hipMemcpyAsync(&bits, op->flag_dev, 4, hipMemcpyDeviceToHost, s));
    DEVCHK(hipStreamSynchronize(s));
    float mx;
    memcpy(&mx, &bits, 4);
    if (!(mx > 0.f) || !std::isfinite(mx)) return 0;                              // all zero / not finite: not this route
    const int sa = (int)std::floor(std::log2((1.0f / mx) * 127.0f));
    if (sa < -64 || sa > 64) return 0;
    if (sa_in) *sa_in = sa;
    // (2) the integer epilogue for this exponent (cached)
    if (sa != op->sa_cached) {
        std::vector<int32_t> bt;
        std::vector<long long> bw;
        if (int rc = y355_op_requant(op->cin, sa, op->e_w, op->e_b, op->act, op->qb.data(), op->cout, op->cout_pad8, &op->rq, &op->frac_bits, &bt, &bw))
            return rc;
        DEVCHK(hipMemcpy(op->bt_dev, bt.data(), sizeof(int) * op->cout_pad8, hipMemcpyHostToDevice));
        DEVCHK(hipMemcpy(op->bw_dev, bw.data(), sizeof(long long) * op->cout_pad8, hipMemcpyHostToDevice));
        op->sa_cached = sa;
    }
    // (3) stage: int8 NHWC with halo + the dyadic verdict
    const size_t in_bytes = ((size_t)batch * (height + 2) * (width + 2) + 64) * op->cpad;
    const size_t raw_bytes = sizeof(long long) * (size_t)batch * height * width * op->cout_pad8;
    const size_t geo = ((size_t)batch << 40) ^ ((size_t)height << 20) ^ (size_t)width ^ ((size_t)1 << 61);
    const bool regeo = geo != op->geo;
    op->geo = geo;
    if (int rc = op_grow(&op->in_dev, &op->in_cap, in_bytes, true)) return rc;
    if (int rc = op_grow(&op->out_dev, &op->out_cap, raw_bytes, false)) return rc;
    if (regeo) DEVCHK(hipMemsetAsync(op->in_dev, 0, in_bytes, s));
    hipLaunchKernelGGL(dyadic_to_nhwc_i8_kernel, dim3(grid_for(n_in)), dim3(256), 0, s, x_dev, (int8_t *)op->in_dev, batch, op->cin, height, width,
                       op->cpad, std::ldexp(1.0f, sa), op->flag_dev + 1);
    DEVCHK(hipGetLastError());
    unsigned int bad = 0;
    DEVCHK(hipMemcpyAsync(&bad, op->flag_dev + 1, 4, hipMemcpyDeviceToHost, s));
    DEVCHK(hipStreamSynchronize(s));
    if (bad) return 0;
    // (4) conv + bias + activation without requantisation (statistics mode dumps t'), (5) t' / 2^F' -> fp32 NCHW
    const int sel = op->cpad == 16 ? 0 : op->cpad == 32 ? 1 : op->cpad == 64 ? 2 : op->cpad == 128 ? 3 : 4;
    const ConvKernelInfo &ki = *y355_conv_kernel(Y355_K_GEN16 + sel);
    DEVCHK(hipMemsetAsync(op->ctr_dev, 0, sizeof(Counters), s));
    ConvParams p{};
    p.in = (const int8_t *)op->in_dev; p.out = nullptr; p.w = op->qw_dev; p.bias_t = op->bt_dev; p.bias_w = op->bw_dev; p.ctr = op->ctr_dev;
    p.raw = (long long *)op->out_dev;
    p.B = batch; p.H = height; p.W = width; p.cstride = op->cout_pad8; p.out_halo = 0;
    p.tiles_x = (width + ki.tw - 1) / ki.tw; p.tiles_y = (height + ki.th - 1) / ki.th; p.nblk = op->cout_pad8 / ki.bn;
    p.rq = op->rq; p.mode = 1; p.guard = 0;
    ki.launch(p, p.tiles_x * p.tiles_y * p.nblk * batch, s);
    DEVCHK(hipGetLastError());
    hipLaunchKernelGGL(raw_to_nchw_f32_kernel, dim3(grid_for(n_out)), dim3(256), 0, s, (const long long *)op->out_dev, out_dev, batch, op->cout, height,
                       width, op->cout_pad8, std::ldexp(1.0f, -op->frac_bits));
    DEVCHK(hipGetLastError());
    *exact = 1;
    return 0;
}

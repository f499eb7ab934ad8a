// yolo355 -- table-driven network executor behind the y355_net_* C ABI (include/yolo355.h):
//   Y355_ARCH_SLIM_V2  models/slim_yolo_v2.py:386-422, forward :549-622  (SlimYOLOv2, the fp32 model)
//   Y355_ARCH_TINY_V3  models/tiny_yolo_v3.py:9-273 + backbone/darknet.py:211-255 (YOLOv3tiny)
// in two arithmetic types: bf16 (BN-folded fp32 weights, bf16 MFMA, fp32 accumulate) and int8
// (per-tensor power-of-two quantisation, the recipe of retune_bias_quantize.py:73-119 applied
// to these graphs).  Every conv is convg.hip; the ops between the convs are the small kernels
// in this file.
#include "../../include/yolo355.h"
#include "y355_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

int y355_fail(int code, const std::string &msg);
int y355_prepare_kernels();
#define HIPCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess)                                                               \
            return y355_fail(Y355_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {
enum { OP_CONV1 = 0, OP_CONV, OP_POOL, OP_UPSAMPLE, OP_INPUT, OP_REORG, OP_SPP };
enum { ACT_NONE = 0, ACT_L125, ACT_L100 };     // LeakyReLU(0.125) utils/modules.py:15; (0.1) backbone/darknet.py:18

struct TensorDef { int C, div, pred; };        // channels; H = height / div; pred: prediction map (no halo)
struct OpDef {
    int type, in, out;
    int choff;        // first channel written in `out` (concat by construction)
    int layer;        // weight slot
    int cin, cout;    // cout 0 = A * (5 + C); cin may be a leading channel range of a wider (concat) buffer
    int ksize, pool, act;
    int stride2;      // 1: 3x3 / pad 1 / stride 2 (backbone/darknet.py:124-141)
    int res1;         // residual tensor + 1 added after the activation (darknet.py:36), 0 = none
};
struct ArchDef { int ntensors; const TensorDef *t; int nops; const OpDef *ops; int nlayers; int nlev; int pred_t[3]; float stride[3]; };

// ---- SlimYOLOv2 (models/slim_yolo_v2.py:403-419, 551-567)
const TensorDef kSlimT[] = {{16, 2, 0}, {32, 4, 0}, {64, 4, 0}, {64, 8, 0}, {128, 8, 0}, {128, 16, 0},
                            {256, 16, 0}, {256, 16, 0}, {256, 16, 0}, {0, 16, 1}};
const OpDef kSlimOps[] = {
    {OP_CONV1, -1, 0, 0, 0, 3, 16, 3, 1, ACT_L125},
    {OP_CONV, 0, 1, 0, 1, 16, 32, 3, 1, ACT_L125},
    {OP_CONV, 1, 2, 0, 2, 32, 64, 3, 0, ACT_L125},
    {OP_CONV, 2, 3, 0, 3, 64, 64, 3, 1, ACT_L125},
    {OP_CONV, 3, 4, 0, 4, 64, 128, 3, 0, ACT_L125},
    {OP_CONV, 4, 5, 0, 5, 128, 128, 3, 1, ACT_L125},
    {OP_CONV, 5, 6, 0, 6, 128, 256, 3, 0, ACT_L125},
    {OP_CONV, 6, 7, 0, 7, 256, 256, 3, 0, ACT_L125},
    {OP_CONV, 7, 8, 0, 8, 256, 256, 3, 0, ACT_L125},
    {OP_CONV, 8, 9, 0, 9, 256, 0, 3, 0, ACT_NONE},
};
// ---- YOLOv3tiny (backbone/darknet.py:215-253, models/tiny_yolo_v3.py:27-39, 176-200)
// tensor 4 is the concat buffer [C_4 (256) | up(conv_1x1_2(C_5)) (128)] (:190)
const TensorDef kTinyT[] = {{16, 2, 0}, {32, 4, 0}, {64, 8, 0}, {128, 16, 0}, {384, 16, 0}, {256, 32, 0}, {512, 32, 0},
                            {512, 32, 0}, {1024, 32, 0}, {256, 32, 0}, {128, 32, 0}, {256, 16, 0}, {512, 32, 0},
                            {0, 16, 1}, {0, 32, 1}};
const OpDef kTinyOps[] = {
    {OP_CONV1, -1, 0, 0, 0, 3, 16, 3, 1, ACT_L100},        // conv_1 + maxpool_1
    {OP_CONV, 0, 1, 0, 1, 16, 32, 3, 1, ACT_L100},         // conv_2 + maxpool_2
    {OP_CONV, 1, 2, 0, 2, 32, 64, 3, 1, ACT_L100},         // conv_3 + maxpool_3
    {OP_CONV, 2, 3, 0, 3, 64, 128, 3, 1, ACT_L100},        // conv_4 + maxpool_4
    {OP_CONV, 3, 4, 0, 4, 128, 256, 3, 0, ACT_L100},       // conv_5 = C_4
    {OP_POOL, 4, 5, 0, -1, 256, 256, 2, 0, 0},             // maxpool_5 (2x2, stride 2)
    {OP_CONV, 5, 6, 0, 5, 256, 512, 3, 0, ACT_L100},       // conv_6
    {OP_POOL, 6, 7, 0, -1, 512, 512, 2, 1, 0},             // maxpool_6: ZeroPad2d((0,1,0,1)) + MaxPool(2, 1)
    {OP_CONV, 7, 8, 0, 6, 512, 1024, 3, 0, ACT_L100},      // conv_7 = C_5
    {OP_CONV, 8, 9, 0, 7, 1024, 256, 3, 0, ACT_L125},      // conv_set_2
    {OP_CONV, 9, 10, 0, 8, 256, 128, 1, 0, ACT_L125},      // conv_1x1_2
    {OP_UPSAMPLE, 10, 4, 256, -1, 128, 128, 0, 0, 0},      // bilinear x2, align_corners (:188)
    {OP_CONV, 4, 11, 0, 9, 384, 256, 3, 0, ACT_L125},      // conv_set_1
    {OP_CONV, 9, 12, 0, 10, 256, 512, 3, 0, ACT_L125},     // extra_conv_2
    {OP_CONV, 12, 14, 0, 11, 512, 0, 1, 0, ACT_NONE},      // pred_2 (stride 32)
    {OP_CONV, 11, 13, 0, 12, 256, 0, 1, 0, ACT_NONE},      // pred_1 (stride 16)
};
// ---- myYOLOv2 (models/yolo_v2.py:26-39, 165-179) on DarkNet-19 (backbone/darknet.py:40-110); bf16 only
const TensorDef kV2T[] = {
    {3, 1, 0},                                                   //  0 input as bf16 NHWC16
    {32, 2, 0}, {64, 4, 0},                                      //  1 conv_1+pool, 2 conv_2+pool
    {128, 4, 0}, {64, 4, 0}, {128, 8, 0},                        //  3..5 conv_3 (last pooled)
    {256, 8, 0}, {128, 8, 0}, {256, 8, 0}, {256, 16, 0},         //  6..8 conv_4 (8 = C_4), 9 maxpool_4
    {512, 16, 0}, {256, 16, 0}, {512, 16, 0}, {256, 16, 0}, {512, 16, 0},   // 10..14 conv_5 (14 = C_5)
    {512, 32, 0},                                                // 15 maxpool_5
    {1024, 32, 0}, {512, 32, 0}, {1024, 32, 0}, {512, 32, 0}, {1024, 32, 0},   // 16..20 conv_6 (20 = C_6)
    {1024, 32, 0},                                               // 21 convsets_1[0]
    {64, 16, 0},                                                 // 22 route_layer
    {1280, 32, 0},                                               // 23 cat(reorg(route) [0:256), convsets_1 [256:1280))
    {1024, 32, 0},                                               // 24 convsets_2
    {0, 32, 1},                                                  // 25 pred
};
const OpDef kV2Ops[] = {
    {OP_INPUT, -1, 0, 0, -1, 3, 3, 0, 0, 0},
    {OP_CONV, 0, 1, 0, 0, 3, 32, 3, 1, ACT_L100},
    {OP_CONV, 1, 2, 0, 1, 32, 64, 3, 1, ACT_L100},
    {OP_CONV, 2, 3, 0, 2, 64, 128, 3, 0, ACT_L100},
    {OP_CONV, 3, 4, 0, 3, 128, 64, 1, 0, ACT_L100},
    {OP_CONV, 4, 5, 0, 4, 64, 128, 3, 1, ACT_L100},
    {OP_CONV, 5, 6, 0, 5, 128, 256, 3, 0, ACT_L100},
    {OP_CONV, 6, 7, 0, 6, 256, 128, 1, 0, ACT_L100},
    {OP_CONV, 7, 8, 0, 7, 128, 256, 3, 0, ACT_L100},
    {OP_POOL, 8, 9, 0, -1, 256, 256, 2, 0, 0},
    {OP_CONV, 9, 10, 0, 8, 256, 512, 3, 0, ACT_L100},
    {OP_CONV, 10, 11, 0, 9, 512, 256, 1, 0, ACT_L100},
    {OP_CONV, 11, 12, 0, 10, 256, 512, 3, 0, ACT_L100},
    {OP_CONV, 12, 13, 0, 11, 512, 256, 1, 0, ACT_L100},
    {OP_CONV, 13, 14, 0, 12, 256, 512, 3, 0, ACT_L100},
    {OP_POOL, 14, 15, 0, -1, 512, 512, 2, 0, 0},
    {OP_CONV, 15, 16, 0, 13, 512, 1024, 3, 0, ACT_L100},
    {OP_CONV, 16, 17, 0, 14, 1024, 512, 1, 0, ACT_L100},
    {OP_CONV, 17, 18, 0, 15, 512, 1024, 3, 0, ACT_L100},
    {OP_CONV, 18, 19, 0, 16, 1024, 512, 1, 0, ACT_L100},
    {OP_CONV, 19, 20, 0, 17, 512, 1024, 3, 0, ACT_L100},
    {OP_CONV, 20, 21, 0, 18, 1024, 1024, 3, 0, ACT_L125},        // convsets_1[0]
    {OP_CONV, 21, 23, 256, 19, 1024, 1024, 3, 0, ACT_L125},      // convsets_1[1] -> cat[256:1280)
    {OP_CONV, 14, 22, 0, 20, 512, 64, 1, 0, ACT_L125},           // route_layer on C_5
    {OP_REORG, 22, 23, 0, -1, 64, 256, 2, 0, 0},                 // reorg(stride 2) -> cat[0:256)
    {OP_CONV, 23, 24, 0, 21, 1280, 1024, 3, 0, ACT_L125},        // convsets_2
    {OP_CONV, 24, 25, 0, 22, 1024, 0, 1, 0, ACT_NONE},           // pred (1x1)
};
// ---- myYOLOv3 / myYOLOv3Spp (models/yolo_v3.py:26-61, 203-231; models/yolo_v3_spp.py:31-36) on DarkNet-53
// (backbone/darknet.py:112-161): built programmatically, weight slots in forward order (bf16 only)
struct V3Graph {
    std::vector<TensorDef> t;
    std::vector<OpDef> ops;
    int nlayers = 0;
    int pred[3] = {0, 0, 0};
    int T(int C, int div, int pred_ = 0) { t.push_back(TensorDef{C, div, pred_}); return (int)t.size() - 1; }
    void conv(int in, int out, int choff, int cin, int cout, int k, int act, int stride2 = 0, int res = -1) {
        ops.push_back(OpDef{OP_CONV, in, out, choff, nlayers++, cin, cout, k, 0, act, stride2, res + 1});
    }
    // resblock(ch) x n on tensor x (div d); the last block may write into `last_out` (a concat buffer, channel offset 0)
    int resblocks(int x, int ch, int d, int n, int last_out = -1) {
        for (int i = 0; i < n; ++i) {
            const int mid = T(ch / 2 < 64 ? 64 : ch / 2, d);                 // >= 64 channels: the kernels write 64-channel blocks
            conv(x, mid, 0, ch, ch / 2, 1, ACT_L100);
            const int out = (i == n - 1 && last_out >= 0) ? last_out : T(ch, d);
            conv(mid, out, 0, ch / 2, ch, 3, ACT_L100, 0, x);
            x = out;
        }
        return x;
    }
    explicit V3Graph(bool spp) {
        const int in = T(3, 1);
        ops.push_back(OpDef{OP_INPUT, -1, in, 0, -1, 3, 3, 0, 0, 0, 0, 0});
        int x = T(64, 1);                                                   // 32 real channels
        conv(in, x, 0, 3, 32, 3, ACT_L100);
        int y = T(64, 2);
        conv(x, y, 0, 32, 64, 3, ACT_L100, 1);
        x = resblocks(y, 64, 2, 1);
        y = T(128, 4); conv(x, y, 0, 64, 128, 3, ACT_L100, 1);
        x = resblocks(y, 128, 4, 2);
        y = T(256, 8); conv(x, y, 0, 128, 256, 3, ACT_L100, 1);
        const int cat1 = T(384, 8);                                          // [C_3 (256) | up(conv_1x1_2) (128)]
        const int c3 = resblocks(y, 256, 8, 8, cat1);
        y = T(512, 16); conv(c3, y, 0, 256, 512, 3, ACT_L100, 1);
        const int cat2 = T(768, 16);                                         // [C_4 (512) | up(conv_1x1_3) (256)]
        const int c4 = resblocks(y, 512, 16, 8, cat2);
        y = T(1024, 32); conv(c4, y, 0, 512, 1024, 3, ACT_L100, 1);
        int c5;
        if (spp) {
            const int sppb = T(4096, 32);                                    // [C_5 | pool5 | pool9 | pool13]
            c5 = resblocks(y, 1024, 32, 4, sppb);
            ops.push_back(OpDef{OP_SPP, c5, c5, 1024, -1, 1024, 3072, 0, 0, 0, 0, 0});
        } else {
            c5 = resblocks(y, 1024, 32, 4);
        }
        // conv_set_3
        int a = T(512, 32); conv(c5, a, 0, spp ? 4096 : 1024, 512, 1, ACT_L125);
        int b = T(1024, 32); conv(a, b, 0, 512, 1024, 3, ACT_L125);
        a = T(512, 32); conv(b, a, 0, 1024, 512, 1, ACT_L125);
        b = T(1024, 32); conv(a, b, 0, 512, 1024, 3, ACT_L125);
        const int f3 = T(512, 32); conv(b, f3, 0, 1024, 512, 1, ACT_L125);
        a = T(256, 32); conv(f3, a, 0, 512, 256, 1, ACT_L125);               // conv_1x1_3
        ops.push_back(OpDef{OP_UPSAMPLE, a, cat2, 512, -1, 256, 256, 0, 0, 0, 0, 0});
        // conv_set_2
        a = T(256, 16); conv(cat2, a, 0, 768, 256, 1, ACT_L125);
        b = T(512, 16); conv(a, b, 0, 256, 512, 3, ACT_L125);
        a = T(256, 16); conv(b, a, 0, 512, 256, 1, ACT_L125);
        b = T(512, 16); conv(a, b, 0, 256, 512, 3, ACT_L125);
        const int f2 = T(256, 16); conv(b, f2, 0, 512, 256, 1, ACT_L125);
        a = T(128, 16); conv(f2, a, 0, 256, 128, 1, ACT_L125);               // conv_1x1_2
        ops.push_back(OpDef{OP_UPSAMPLE, a, cat1, 256, -1, 128, 128, 0, 0, 0, 0, 0});
        // conv_set_1
        a = T(128, 8); conv(cat1, a, 0, 384, 128, 1, ACT_L125);
        b = T(256, 8); conv(a, b, 0, 128, 256, 3, ACT_L125);
        a = T(128, 8); conv(b, a, 0, 256, 128, 1, ACT_L125);
        b = T(256, 8); conv(a, b, 0, 128, 256, 3, ACT_L125);
        const int f1 = T(128, 8); conv(b, f1, 0, 256, 128, 1, ACT_L125);
        // heads: extra_conv_3 + pred_3, extra_conv_2 + pred_2, extra_conv_1 + pred_1 (models/yolo_v3.py:219-231)
        a = T(1024, 32); conv(f3, a, 0, 512, 1024, 3, ACT_L125);
        pred[2] = T(0, 32, 1); conv(a, pred[2], 0, 1024, 0, 1, ACT_NONE);
        a = T(512, 16); conv(f2, a, 0, 256, 512, 3, ACT_L125);
        pred[1] = T(0, 16, 1); conv(a, pred[1], 0, 512, 0, 1, ACT_NONE);
        a = T(256, 8); conv(f1, a, 0, 128, 256, 3, ACT_L125);
        pred[0] = T(0, 8, 1); conv(a, pred[0], 0, 256, 0, 1, ACT_NONE);
    }
    ArchDef arch() const {
        return ArchDef{(int)t.size(), t.data(), (int)ops.size(), ops.data(), nlayers, 3, {pred[0], pred[1], pred[2]}, {8.f, 16.f, 32.f}};
    }
};
const V3Graph kV3(false), kV3Spp(true);
const ArchDef kArch[5] = {
    {10, kSlimT, 10, kSlimOps, 10, 1, {9, -1, -1}, {16.f, 0.f, 0.f}},
    {15, kTinyT, 16, kTinyOps, 13, 2, {13, 14, -1}, {16.f, 32.f, 0.f}},
    {26, kV2T, 27, kV2Ops, 23, 1, {25, -1, -1}, {32.f, 0.f, 0.f}},
    kV3.arch(),
    kV3Spp.arch(),
};

struct Tensor {
    int C = 0, Cpad = 0, H = 0, W = 0, halo = 1, pred = 0;
    size_t pb = 0;          // bytes per pixel
    char *dev = nullptr;
    size_t bytes = 0;
};
struct NLayer {
    int cin = 0, cout = 0, cout_pad = 0, ksize = 3, kid = -1, op = -1;
    bool loaded = false;
    char *w_dev = nullptr;
    float *bias_dev = nullptr;
    size_t w_bytes = 0;
    // int8 nets
    std::vector<int32_t> q_b;
    int e_w = 0, e_b = 0;
    long long *bias_w_dev = nullptr;
    RequantG rq{};
    Requant rq1{};            // first layer (conv1.hip epilogue)
    bool dirty = true;
    // 3x3 layers that have a ring instantiation (convr.hip): its id and the weights in its fragment order
    long long wabs = 0;       // int8: max over output channels of sum |q_w| (0 = unknown)
    int rid = -1;
    char *wr_dev = nullptr;
    size_t wr_bytes = 0;
    // bf16 thin 3x3 layers with the weights in registers (convpxb.hip): its id and the weights in its fragment order
    int pbid = -1;
    char *wpb_dev = nullptr;
};

__global__ void pool_bf16_kernel(const char *in, char *out, int B, int Hin, int Win, int in_pb, int cbytes, int Ho, int Wo,
                                 int out_pb, int stride) {
    // 16 bytes (8 bf16 channels) of one output pixel per thread; the zero halo IS the padding
    const int cg = cbytes / 16;
    const size_t total = (size_t)B * Ho * Wo * cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cg);
        size_t r = i / cg;
        const int x = (int)(r % Wo);
        r /= Wo;
        const int y = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const char *src = in + (((size_t)b * (Hin + 2) + y * stride + 1) * (Win + 2) + x * stride + 1) * in_pb + c * 16;
        uint4 v[4];
        v[0] = *(const uint4 *)src;
        v[1] = *(const uint4 *)(src + in_pb);
        v[2] = *(const uint4 *)(src + (size_t)(Win + 2) * in_pb);
        v[3] = *(const uint4 *)(src + (size_t)(Win + 2) * in_pb + in_pb);
        unsigned int o[4];
        const unsigned int *w0 = (const unsigned int *)&v[0], *w1 = (const unsigned int *)&v[1];
        const unsigned int *w2 = (const unsigned int *)&v[2], *w3 = (const unsigned int *)&v[3];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned int res = 0;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const int sh = 16 * hf;
                const float a = __uint_as_float(((w0[k] >> sh) & 0xffffu) << 16), bb = __uint_as_float(((w1[k] >> sh) & 0xffffu) << 16);
                const float cc = __uint_as_float(((w2[k] >> sh) & 0xffffu) << 16), d = __uint_as_float(((w3[k] >> sh) & 0xffffu) << 16);
                const float m = fmaxf(fmaxf(a, bb), fmaxf(cc, d));
                res |= (__float_as_uint(m) >> 16) << sh;
            }
            o[k] = res;
        }
        *(uint4 *)(out + (((size_t)b * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * out_pb + c * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// F.interpolate(scale_factor=2, mode='bilinear', align_corners=True) (models/tiny_yolo_v3.py:188):
// src = dst * (in - 1) / (out - 1), the two-tap blend of torch's upsample_bilinear2d in fp32.
__global__ void upsample_bf16_kernel(const char *in, char *out, int B, int Hin, int Win, int in_pb, int C, int out_pb,
                                     int out_off, float ry, float rx) {
    const int Ho = 2 * Hin, Wo = 2 * Win;
    const size_t total = (size_t)B * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int x = (int)(r % Wo);
        r /= Wo;
        const int y = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const float sy = ry * (float)y, sx = rx * (float)x;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = min(y0 + 1, Hin - 1), x1 = min(x0 + 1, Win - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        auto ld = [&](int yy, int xx) -> float {
            const unsigned short h = *(const unsigned short *)(in + (((size_t)b * (Hin + 2) + yy + 1) * (Win + 2) + xx + 1) * in_pb + c * 2);
            return __uint_as_float((unsigned int)h << 16);
        };
        const float v = hy * (hx * ld(y0, x0) + lx * ld(y0, x1)) + ly * (hx * ld(y1, x0) + lx * ld(y1, x1));
        *(unsigned short *)(out + (((size_t)b * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * out_pb + out_off + c * 2) =
            __builtin_bit_cast(unsigned short, (__bf16)v);
    }
}

__global__ void pool_i8_kernel(const char *in, char *out, int B, int Hin, int Win, int in_pb, int cbytes, int Ho, int Wo,
                               int out_pb, int stride) {
    const int cg = cbytes / 16;
    const size_t total = (size_t)B * Ho * Wo * cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % cg);
        size_t r = i / cg;
        const int x = (int)(r % Wo);
        r /= Wo;
        const int y = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const char *src = in + (((size_t)b * (Hin + 2) + y * stride + 1) * (Win + 2) + x * stride + 1) * in_pb + c * 16;
        uint4 v[4];
        v[0] = *(const uint4 *)src;
        v[1] = *(const uint4 *)(src + in_pb);
        v[2] = *(const uint4 *)(src + (size_t)(Win + 2) * in_pb);
        v[3] = *(const uint4 *)(src + (size_t)(Win + 2) * in_pb + in_pb);
        unsigned int o[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            unsigned int res = 0;
#pragma unroll
            for (int by = 0; by < 4; ++by) {
                int m = -128;
#pragma unroll
                for (int j = 0; j < 4; ++j) m = max(m, (int)(signed char)((((const unsigned int *)&v[j])[k] >> (8 * by)) & 0xffu));
                res |= (unsigned int)(m & 0xff) << (8 * by);
            }
            o[k] = res;
        }
        *(uint4 *)(out + (((size_t)b * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * out_pb + c * 16) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// int8 form of the bilinear x2: the blend of the integer values in fp32 (same expression as the bf16
// kernel), rescaled by the power of two between the two tensors' exponents, rounded half-to-even.
__global__ void upsample_i8_kernel(const char *in, char *out, int B, int Hin, int Win, int in_pb, int C, int out_pb,
                                   int out_off, float ry, float rx, float rescale) {
    const int Ho = 2 * Hin, Wo = 2 * Win;
    const size_t total = (size_t)B * Ho * Wo * C;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t r = i / C;
        const int x = (int)(r % Wo);
        r /= Wo;
        const int y = (int)(r % Ho);
        const int b = (int)(r / Ho);
        const float sy = ry * (float)y, sx = rx * (float)x;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = min(y0 + 1, Hin - 1), x1 = min(x0 + 1, Win - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        auto ld = [&](int yy, int xx) -> float {
            return (float)*(const signed char *)(in + (((size_t)b * (Hin + 2) + yy + 1) * (Win + 2) + xx + 1) * in_pb + c);
        };
        const float v = hy * (hx * ld(y0, x0) + lx * ld(y0, x1)) + ly * (hx * ld(y1, x0) + lx * ld(y1, x1));
        const float q = fminf(fmaxf(rintf(v * rescale), -127.f), 127.f);
        *(signed char *)(out + (((size_t)b * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * out_pb + out_off + c) = (signed char)(int)q;
    }
}

// the same, sixteen channels of one output pixel per thread (16-byte loads and stores; C % 16 == 0, 16-byte aligned pixels):
// element for element the expression above
__global__ void upsample_i8x16_kernel(const char *in, char *out, int B, int Hin, int Win, int in_pb, int C, int out_pb,
                                      int out_off, float ry, float rx, float rescale) {
    const int Ho = 2 * Hin, Wo = 2 * Win, CG = C / 16;
    const int total = B * Ho * Wo * CG;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int cg = i % CG;
        int r = i / CG;
        const int x = r % Wo;
        r /= Wo;
        const int y = r % Ho;
        const int b = r / Ho;
        const float sy = ry * (float)y, sx = rx * (float)x;
        const int y0 = (int)sy, x0 = (int)sx;
        const int y1 = min(y0 + 1, Hin - 1), x1 = min(x0 + 1, Win - 1);
        const float ly = sy - (float)y0, lx = sx - (float)x0;
        const float hy = 1.f - ly, hx = 1.f - lx;
        auto ld = [&](int yy, int xx) { return *(const v4i *)(in + (((size_t)b * (Hin + 2) + yy + 1) * (Win + 2) + xx + 1) * in_pb + cg * 16); };
        const v4i a00 = ld(y0, x0), a01 = ld(y0, x1), a10 = ld(y1, x0), a11 = ld(y1, x1);
        v4i o;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            unsigned int word = 0;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float f00 = (float)(signed char)(a00[w] >> (8 * k)), f01 = (float)(signed char)(a01[w] >> (8 * k));
                const float f10 = (float)(signed char)(a10[w] >> (8 * k)), f11 = (float)(signed char)(a11[w] >> (8 * k));
                const float v = hy * (hx * f00 + lx * f01) + ly * (hx * f10 + lx * f11);
                const float q = fminf(fmaxf(rintf(v * rescale), -127.f), 127.f);
                word |= ((unsigned int)(int)q & 0xffu) << (8 * k);
            }
            o[w] = (int)word;
        }
        *(v4i *)(out + (((size_t)b * (Ho + 2) + y + 1) * (Wo + 2) + x + 1) * out_pb + out_off + cg * 16) = o;
    }
}

__global__ void absmax_bf16_kernel(const char *t, size_t n_elems, unsigned int *out) {
    float m = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n_elems; i += (size_t)gridDim.x * blockDim.x) {
        const unsigned short h = ((const unsigned short *)t)[i];
        m = fmaxf(m, fabsf(__uint_as_float((unsigned int)h << 16)));
    }
    const unsigned int u = y355_wave_max_u32(__float_as_uint(m));
    if ((threadIdx.x & 63) == 0) atomicMax(out, u);
}
}  // namespace

#ifndef Y355_USE_CONVPXB
#define Y355_USE_CONVPXB 1          // 0 (A/B builds): no convpxb.hip instantiation is ever selected
#endif
#ifndef Y355_USE_CONVR
#define Y355_USE_CONVR 1            // 0 (A/B builds): every layer of the generic nets on convg.hip
#endif
struct y355_net {
    y355_net_config cfg{};
    const ArchDef *arch = nullptr;
    bool bf = true;
    int es = 2;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    std::vector<Tensor> T;
    std::vector<NLayer> L;
    char *w0_dev = nullptr;           // first-layer fragments
    int predc = 0, N = 0, max_det = 0;
    y355_head_ws ws{};
    float *cand_box = nullptr, *cand_score = nullptr;
    int *cand_cls = nullptr;
    unsigned int *absmax_dev = nullptr;
    // int8 nets: activation exponents (value = q / 2^sa) of the input and of every tensor
    int sa_in = 0;
    bool sa_ok = false;
    std::vector<int> sa;
    Counters *ctr_dev = nullptr;      // [nops + 1]: the set the last forward counted into (one of ctrs' two)
    CounterSets ctrs;
    // int8 nets whose graph starts with conv(3 -> 16) + pool, conv(16 -> 32) + pool (SlimYOLOv2, YOLOv3tiny:
    // models/slim_yolo_v2.py:549-575, backbone/darknet.py:216-220): both layers in ONE launch of the q_bf engine's fused front
    // end (front.hip) when their epilogues are exact in fp32 (y355_front_eligible); tap forwards run the layers one by one
    bool front_graph = false, front_ok = false, front_dirty = true;
    bool t0_skipped = false;                            // the last forward ran the fused front end: tensor 0 (conv1's map) was not written
    int8_t *wf_dev = nullptr;         // 16 KiB of front-end weight fragments (y355_pack_front)
    int *fb1_dev = nullptr, *fb2_dev = nullptr;   // pre-shifted int32 biases of the two layers
    Requant frq1{}, frq2{};
    int no_pxb = 0;                   // Y355_NET_OPT_THIN_RESIDENT = 0: the thin 3x3 layers of the bf16 graphs on convr.hip instead of convpxb.hip
    int tput_wgs = 0;                 // Y355_NET_OPT_WORKGROUPS: persistent workgroups per convr launch while several handles share the GPU (0 = one per CU)
    int profile = 0;
    std::vector<hipEvent_t> ev;
    std::vector<void *> allocs;
};

static int in_kbytes(const y355_net *h, const OpDef &o);

static int nmalloc(y355_net *h, void **p, size_t bytes, bool zero) {
    HIPCHK(hipMalloc(p, bytes ? bytes : 16));
    h->allocs.push_back(*p);
    if (zero) HIPCHK(hipMemset(*p, 0, bytes ? bytes : 16));
    return 0;
}

extern "C" void y355_net_destroy(y355_net *h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device_id);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) (void)hipFree(p);
    for (auto &e : h->ev) (void)hipEventDestroy(e);
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int y355_net_create(const y355_net_config *cfg, y355_net **out) {
    if (!cfg || !out) return y355_fail(Y355_EINVAL, "null argument");
    if (cfg->arch < 0 || cfg->arch > Y355_ARCH_YOLO_V3_SPP) return y355_fail(Y355_EINVAL, "unknown arch");
    if (cfg->arch >= Y355_ARCH_YOLO_V2 && cfg->dtype != Y355_DT_BF16)
        return y355_fail(Y355_EINVAL, "yolo_v2 / yolo_v3 / yolo_v3_spp are built in bf16 only (the reference has no quantized form of them)");
    if (cfg->dtype != Y355_DT_BF16 && cfg->dtype != Y355_DT_INT8) return y355_fail(Y355_EINVAL, "unknown dtype");
    // SlimYOLOv2 has four 2x2 pools (stride 16: models/slim_yolo_v2.py:52); the other graphs reach stride 32
    const int mult = cfg->arch == Y355_ARCH_SLIM_V2 ? 16 : 32;
    if (cfg->height <= 0 || cfg->width <= 0 || cfg->height % mult || cfg->width % mult)
        return y355_fail(Y355_EINVAL, cfg->arch == Y355_ARCH_SLIM_V2 ? "input size must be a positive multiple of 16"
                                                                      : "input size must be a positive multiple of 32");
    const ArchDef &A = kArch[cfg->arch];
    if (cfg->num_anchors < 1 || cfg->num_anchors * A.nlev > Y355_HEAD_MAXA || cfg->num_classes < 1)
        return y355_fail(Y355_EINVAL, "bad anchors / classes");
    if (cfg->max_batch < 1) return y355_fail(Y355_EINVAL, "max_batch < 1");
    const int predc = cfg->num_anchors * (5 + cfg->num_classes);
    if (predc > 256) return y355_fail(Y355_EINVAL, "A*(5+C) > 256 not supported");
    int N = 0;
    for (int l = 0; l < A.nlev; ++l) N += (cfg->height / (int)A.stride[l]) * (cfg->width / (int)A.stride[l]) * cfg->num_anchors;
    const int Hb = cfg->height / (int)A.stride[0], Wb = cfg->width / (int)A.stride[0];
    (void)Hb;
    (void)Wb;
    if (N > Y355_NMS_CAP && A.nlev < 3)
        return y355_fail(Y355_EINVAL, "more than 4096 anchors / sort bins per image not supported");
    if (N > 16 * Y355_NMS_CAP) return y355_fail(Y355_EINVAL, "more than 65536 anchors per image not supported");
    HIPCHK(hipSetDevice(cfg->device_id));
    if (int e = y355_prepare_kernels()) return e;
    if (y355_prepare_convr(cfg->device_id)) return y355_fail(Y355_EHIP, "hipFuncSetAttribute(convr) / sink allocation failed");
    if (y355_prepare_convpxb()) return y355_fail(Y355_EHIP, "hipFuncSetAttribute(convpxb) failed");
    y355_net *h = new y355_net();
    h->cfg = *cfg;
    h->arch = &A;
    h->bf = cfg->dtype == Y355_DT_BF16;
    h->es = h->bf ? 2 : 1;
    h->predc = predc;
    h->sa.assign(A.ntensors, 0);
    h->N = N;
    const int ncand = N > Y355_NMS_CAP ? Y355_NMS_CAP : N;     // larger heads are thresholded and compacted first
    h->max_det = (cfg->max_det <= 0 || cfg->max_det > ncand) ? ncand : cfg->max_det;
    if (!cfg->own_stream) h->stream = (hipStream_t)cfg->stream;
    else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
            delete h;
            return y355_fail(Y355_EHIP, "hipStreamCreate failed");
        }
        h->own_stream = true;
    }
    const int B = cfg->max_batch;
    int rc = 0;
    h->T.resize(A.ntensors);
    for (int i = 0; i < A.ntensors && !rc; ++i) {
        Tensor &t = h->T[i];
        t.C = A.t[i].C ? A.t[i].C : predc;
        t.pred = A.t[i].pred;
        t.halo = t.pred ? 0 : 1;
        t.H = cfg->height / A.t[i].div;
        t.W = cfg->width / A.t[i].div;
        // a pixel is at least 32 bytes (the thin path of convg.hip)
        const int cq = h->bf ? 16 : 32;
        t.Cpad = t.pred ? (t.C <= 64 ? 64 : t.C <= 128 ? 128 : 256) : (t.C + cq - 1) / cq * cq;
        t.pb = (size_t)t.Cpad * ((t.pred && h->bf) ? 4 : h->es);
        t.bytes = ((size_t)B * (t.H + 2 * t.halo) * (t.W + 2 * t.halo) + 64) * t.pb;
        rc = nmalloc(h, (void **)&t.dev, t.bytes, true);
    }
    h->L.resize(A.nlayers);
    for (int i = 0; i < A.nops && !rc; ++i) {
        const OpDef &o = A.ops[i];
        if (o.type != OP_CONV1 && o.type != OP_CONV) continue;
        NLayer &L = h->L[o.layer];
        L.op = i;
        L.cin = o.cin;
        L.cout = o.cout ? o.cout : predc;
        L.ksize = o.ksize;
        if (o.type == OP_CONV1) {
            L.cout_pad = 16;
            rc = nmalloc(h, (void **)&h->w0_dev, 2048, true);
        } else {
            const Tensor &ti = h->T[o.in];
            L.kid = y355_convg_select(in_kbytes(h, o), L.cout, o.pool, ti.H, ti.W, o.stride2 ? 2 : 1, h->cfg.max_batch);
            if (L.kid < 0) { rc = y355_fail(Y355_EINVAL, "no convolution kernel for a layer of this graph"); break; }
            const ConvGInfo &ki = *y355_convg_kernel(h->bf, L.kid);
            if (in_kbytes(h, o) % ki.chb) { rc = y355_fail(Y355_EINVAL, "input channels not a multiple of the kernel's chunk"); break; }
            L.cout_pad = (L.cout + ki.bn - 1) / ki.bn * ki.bn;
            if (!h->T[o.out].pred && o.choff + L.cout_pad > h->T[o.out].Cpad) { rc = y355_fail(Y355_EINVAL, "layer wider than its output buffer"); break; }
            if (h->T[o.out].pred && L.cout_pad > h->T[o.out].Cpad) {
                rc = y355_fail(Y355_EINVAL, "prediction map wider than its buffer");
                break;
            }
            L.w_bytes = y355_convg_packed_bytes(ki, in_kbytes(h, o), o.ksize * o.ksize, L.cout_pad);
            rc = nmalloc(h, (void **)&L.w_dev, L.w_bytes, true);
            // (the pointwise launcher derives its k-steps from the input buffer's pixel pitch and has no pool: only an op that
            // consumes the WHOLE buffer, unpooled, may take it -- ADVICE r3)
            if (!rc && !h->bf && o.ksize == 1 && !o.stride2 && !o.res1 && !o.pool && ti.pb == in_kbytes(h, o) && L.cout_pad % 64 == 0 &&
                in_kbytes(h, o) % 64 == 0 && in_kbytes(h, o) <= 1024 && Y355_USE_CONVR) {                              // int8 1x1: pointwise kernel of convr.hip (when the epilogue fits 32 bits)
                L.rid = -2;
                L.wr_bytes = (size_t)(in_kbytes(h, o) / 64) * (L.cout_pad / 16) * 1024;
                rc = nmalloc(h, (void **)&L.wr_dev, L.wr_bytes, true);
            }
            if (!rc && h->bf && o.ksize == 3 && !o.stride2 && !o.res1 && !h->T[o.out].pred && Y355_USE_CONVPXB) {
                L.pbid = y355_convpxb_select(in_kbytes(h, o), L.cout, o.pool);
                if (L.pbid >= 0 && ((int)ti.pb != in_kbytes(h, o) || L.cout_pad != L.cout)) L.pbid = -1;     // whole-pixel inputs only
                if (L.pbid >= 0) rc = nmalloc(h, (void **)&L.wpb_dev, y355_convpxb_packed_bytes(L.pbid), true);
            }
            if (!rc && o.ksize == 3 && !o.stride2 && !o.res1 && Y355_USE_CONVR) {
                L.rid = y355_convr_select(h->bf ? 1 : 0, in_kbytes(h, o), L.cout_pad, o.pool, ti.H, ti.W);
                if (L.rid >= 0) {
                    L.wr_bytes = (size_t)(in_kbytes(h, o) / 64) * 9 * (L.cout_pad / 16) * 1024;
                    rc = nmalloc(h, (void **)&L.wr_dev, L.wr_bytes, true);
                }
            }
        }
        if (!rc) rc = nmalloc(h, (void **)&L.bias_dev, sizeof(float) * L.cout_pad, true);
        if (!rc) rc = nmalloc(h, (void **)&L.bias_w_dev, sizeof(long long) * L.cout_pad, true);
    }
    if (!rc && A.nops >= 2) {
        const OpDef &o0 = A.ops[0], &o1 = A.ops[1];
        bool fg = o0.type == OP_CONV1 && o0.cout == 16 && o0.pool == 1 && o1.type == OP_CONV && o1.in == o0.out && o1.cin == 16 &&
                  o1.cout == 32 && o1.ksize == 3 && o1.pool == 1 && !o1.stride2 && !o1.res1 && o1.choff == 0 &&
                  h->T[o1.out].pb == (size_t)(32 * h->es) && h->T[o1.out].halo == 1 && !h->T[o1.out].pred && cfg->height % 4 == 0 &&
                  cfg->width % 4 == 0;
        for (int i = 2; fg && i < A.nops; ++i)                     // nobody else reads conv1's map
            if (A.ops[i].in == o0.out || A.ops[i].res1 - 1 == o0.out) fg = false;
        if (fg) {
            h->front_graph = true;
            rc = nmalloc(h, (void **)&h->wf_dev, h->bf ? 32768 : 16384, true);      // y355_pack_frontb / y355_pack_front
            if (!rc && !h->bf) rc = nmalloc(h, (void **)&h->fb1_dev, sizeof(int) * 16, true);
            if (!rc && !h->bf) rc = nmalloc(h, (void **)&h->fb2_dev, sizeof(int) * 32, true);
        }
    }
    const size_t cap = Y355_NMS_CAP;
    if (!rc) rc = nmalloc(h, (void **)&h->absmax_dev, 16, true);
    if (!rc) rc = nmalloc(h, (void **)&h->ctr_dev, sizeof(Counters) * 2 * (A.nops + 1), true);
    h->ctrs.base = h->ctr_dev;
    h->ctrs.n = A.nops + 1;
    h->ctrs.clean[0] = h->ctrs.clean[1] = true;
    if (!rc) rc = nmalloc(h, &h->ws.cbox, sizeof(float) * 4 * cap * B, false);
    if (!rc) rc = nmalloc(h, &h->ws.cscore, sizeof(float) * cap * B, false);
    if (!rc) rc = nmalloc(h, &h->ws.ccls, sizeof(int) * cap * B, false);
    if (!rc) rc = nmalloc(h, &h->ws.corig, sizeof(int) * cap * B, false);
    if (!rc) rc = nmalloc(h, &h->ws.count, sizeof(int) * B, true);
    if (!rc) rc = nmalloc(h, &h->ws.edges, sizeof(unsigned int) * (size_t)Y355_HEAD_EDGE_CAP * B, false);      // EDGE_CAP pairs per image
    if (!rc) rc = nmalloc(h, &h->ws.nedges, sizeof(int) * 2 * (size_t)B, true);
    if (!rc) rc = nmalloc(h, &h->ws.binstart, sizeof(int) * (cap + 8) * B, true);
    if (!rc) rc = nmalloc(h, &h->ws.astat, sizeof(float) * 4 * Y355_HEAD_MAXG * B, true);
    if (!rc) rc = nmalloc(h, &h->ws.tiny, sizeof(int) * cap * B, true);
    if (!rc) rc = nmalloc(h, &h->ws.ntiny, sizeof(int) * B, true);
    if (!rc) rc = nmalloc(h, &h->ws.dbox, sizeof(float) * 4 * cap * B, true);
    if (!rc) rc = nmalloc(h, &h->ws.dscore, sizeof(float) * cap * B, true);
    if (!rc) rc = nmalloc(h, &h->ws.dcls, sizeof(int) * cap * B, true);
    if (!rc) rc = nmalloc(h, &h->ws.ctype, sizeof(int) * cap * B, true);          // candidate groups
    if (N > Y355_NMS_CAP) {
        h->ws.rstride = (N + 3) / 4 * 4;
        if (!rc) rc = nmalloc(h, &h->ws.rbox, sizeof(float) * 4 * (size_t)h->ws.rstride * B, true);
        if (!rc) rc = nmalloc(h, &h->ws.rscore, sizeof(float) * (size_t)h->ws.rstride * B, true);
        if (!rc) rc = nmalloc(h, &h->ws.rcls, sizeof(int) * (size_t)h->ws.rstride * B, true);
        if (!rc) rc = nmalloc(h, &h->ws.rcount, sizeof(int) * B, true);
        if (!rc) rc = nmalloc(h, &h->ws.ovf, sizeof(int) * B, true);
    }
    if (!rc) rc = nmalloc(h, (void **)&h->cand_box, sizeof(float) * 4 * N * B, false);
    if (!rc) rc = nmalloc(h, (void **)&h->cand_score, sizeof(float) * N * B, false);
    if (!rc) rc = nmalloc(h, (void **)&h->cand_cls, sizeof(int) * N * B, false);
    if (!rc) {
        h->ev.resize(A.nops + 3);
        for (auto &e : h->ev)
            if (hipEventCreate(&e) != hipSuccess) { rc = y355_fail(Y355_EHIP, "hipEventCreate failed"); break; }
    }
    if (rc) {
        std::string keep = y355_last_error();
        y355_net_destroy(h);
        y355_fail(rc, keep);
        return rc;
    }
    *out = h;
    return 0;
}

extern "C" int y355_net_set_option(y355_net *h, int option, int value) {
    if (!h) return y355_fail(Y355_EINVAL, "null net");
    if (option == Y355_NET_OPT_WORKGROUPS) {
        if (value < 0 || value > 256) return y355_fail(Y355_EINVAL, "workgroups per launch: 0 .. 256");
        h->tput_wgs = value;
        return 0;
    }
    if (option == Y355_NET_OPT_THIN_RESIDENT) {
        if (value != 0 && value != 1) return y355_fail(Y355_EINVAL, "Y355_NET_OPT_THIN_RESIDENT takes 0 or 1");
        h->no_pxb = !value;
        return 0;
    }
    return y355_fail(Y355_EINVAL, "unknown option");
}

extern "C" int y355_net_set_thresholds(y355_net *h, float conf, float nms) {
    if (!h) return y355_fail(Y355_EINVAL, "null net");
    h->cfg.conf_thresh = conf;
    h->cfg.nms_thresh = nms;
    return 0;
}

extern "C" int y355_net_num_layers(y355_net *h) { return h ? h->arch->nlayers : Y355_EINVAL; }
extern "C" int y355_net_num_tensors(y355_net *h) { return h ? h->arch->ntensors : Y355_EINVAL; }
extern "C" int y355_net_max_det(y355_net *h) { return h ? h->max_det : Y355_EINVAL; }
extern "C" int y355_net_num_anchors_total(y355_net *h) { return h ? h->N : Y355_EINVAL; }

extern "C" int y355_net_layer_shape(y355_net *h, int idx, int32_t *shape) {
    if (!h || !shape || idx < 0 || idx >= h->arch->nlayers) return y355_fail(Y355_EINVAL, "bad argument");
    const NLayer &L = h->L[idx];
    shape[0] = L.cout; shape[1] = L.cin; shape[2] = L.ksize; shape[3] = L.ksize;
    return 0;
}

extern "C" int y355_net_tensor_shape(y355_net *h, int idx, int32_t *shape) {
    if (!h || !shape || idx < 0 || idx >= h->arch->ntensors) return y355_fail(Y355_EINVAL, "bad argument");
    const Tensor &t = h->T[idx];
    shape[0] = t.C; shape[1] = t.H; shape[2] = t.W;
    return 0;
}

extern "C" int y355_net_load_layer_f32(y355_net *h, int idx, const float *w, const float *b, int cout, int cin, int ksize) {
    if (!h || !w) return y355_fail(Y355_EINVAL, "null argument");
    if (idx < 0 || idx >= h->arch->nlayers) return y355_fail(Y355_EINVAL, "layer index out of range");
    if (!h->bf) return y355_fail(Y355_EINVAL, "fp32 weights go to a bf16 net");
    NLayer &L = h->L[idx];
    if (cout != L.cout || cin != L.cin || ksize != L.ksize) {
        char buf[160];
        snprintf(buf, sizeof buf, "layer %d expects [%d,%d,%d,%d], got [%d,%d,%d,%d]", idx, L.cout, L.cin, L.ksize, L.ksize,
                 cout, cin, ksize, ksize);
        return y355_fail(Y355_EINVAL, buf);
    }
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    const OpDef &o = h->arch->ops[L.op];
    if (o.type == OP_CONV1) {
        char frag[2048];
        y355_pack_conv1f(w, frag);
        HIPCHK(hipMemcpy(h->w0_dev, frag, 2048, hipMemcpyHostToDevice));
    } else {
        const ConvGInfo &ki = *y355_convg_kernel(1, L.kid);
        std::vector<char> packed(L.w_bytes);
        y355_convg_pack(ki, w, nullptr, cout, cin, ksize, in_kbytes(h, o), L.cout_pad, packed.data());
        HIPCHK(hipMemcpy(L.w_dev, packed.data(), packed.size(), hipMemcpyHostToDevice));
        if (L.rid >= 0) {                                      // the same fragments in the ring kernel's (bn, wn, nt) order
            const Y355ConvRInfo &ri = *y355_convr_info(L.rid);
            ConvGInfo kr{};
            kr.bf = 1; kr.chb = 64; kr.bn = ri.bn; kr.wn = ri.wn; kr.nt = ri.nt;
            std::vector<char> pr(L.wr_bytes);
            y355_convg_pack(kr, w, nullptr, cout, cin, ksize, in_kbytes(h, o), L.cout_pad, pr.data());
            HIPCHK(hipMemcpy(L.wr_dev, pr.data(), pr.size(), hipMemcpyHostToDevice));
        }
        if (L.pbid >= 0) {                                     // ... and in convpxb.hip's (weights stay in registers)
            std::vector<char> pb(y355_convpxb_packed_bytes(L.pbid));
            if (y355_convpxb_pack(L.pbid, w, cout, cin, pb.data())) HIPCHK(hipMemcpy(L.wpb_dev, pb.data(), pb.size(), hipMemcpyHostToDevice));
            else L.pbid = -1;
        }
    }
    if (h->front_graph && (L.op == 0 || L.op == 1)) {         // the same weights as the bf16 front end's fragments (frontb.hip)
        std::vector<char> wf(32768);
        y355_pack_frontb(L.op == 0 ? w : nullptr, L.op == 1 ? w : nullptr, wf.data());
        if (L.op == 0) HIPCHK(hipMemcpy(h->wf_dev, wf.data(), 8192, hipMemcpyHostToDevice));
        else HIPCHK(hipMemcpy(h->wf_dev + 8192, wf.data() + 8192, 32768 - 8192, hipMemcpyHostToDevice));
    }
    std::vector<float> bias(L.cout_pad, 0.f);
    if (b) memcpy(bias.data(), b, sizeof(float) * cout);
    HIPCHK(hipMemcpy(L.bias_dev, bias.data(), sizeof(float) * L.cout_pad, hipMemcpyHostToDevice));
    L.loaded = true;
    return 0;
}

// replaces load_state_dict of a checkpoint written by quantize_layers (retune_bias_quantize.py:111-119)
// for one conv of an int8 net: q_w int8 [cout][cin][k][k] (value q_w / 2^e_w), q_b int32 [cout]
extern "C" int y355_net_load_layer_i8(y355_net *h, int idx, const int8_t *q_w, const int32_t *q_b, int cout, int cin,
                                      int ksize, int e_w, int e_b) {
    if (!h || !q_w || !q_b) return y355_fail(Y355_EINVAL, "null argument");
    if (idx < 0 || idx >= h->arch->nlayers) return y355_fail(Y355_EINVAL, "layer index out of range");
    if (h->bf) return y355_fail(Y355_EINVAL, "int8 weights go to an int8 net");
    NLayer &L = h->L[idx];
    if (cout != L.cout || cin != L.cin || ksize != L.ksize) {
        char buf[160];
        snprintf(buf, sizeof buf, "layer %d expects [%d,%d,%d,%d], got [%d,%d,%d,%d]", idx, L.cout, L.cin, L.ksize, L.ksize,
                 cout, cin, ksize, ksize);
        return y355_fail(Y355_EINVAL, buf);
    }
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    const OpDef &o = h->arch->ops[L.op];
    if (o.type == OP_CONV1) {
        int8_t frag[1024];
        y355_pack_conv1(q_w, frag);
        HIPCHK(hipMemcpy(h->w0_dev, frag, 1024, hipMemcpyHostToDevice));
    } else {
        const ConvGInfo &ki = *y355_convg_kernel(0, L.kid);
        std::vector<char> packed(L.w_bytes);
        y355_convg_pack(ki, nullptr, q_w, cout, cin, ksize, in_kbytes(h, o), L.cout_pad, packed.data());
        HIPCHK(hipMemcpy(L.w_dev, packed.data(), packed.size(), hipMemcpyHostToDevice));
        if (L.rid >= 0 || L.rid == -2) {                       // the same fragments in the ring / pointwise kernel's (bn, wn, nt) order
            ConvGInfo kr{};
            kr.bf = 0; kr.chb = 64; kr.bn = 64; kr.wn = 1; kr.nt = 4;
            if (L.rid >= 0) {
                const Y355ConvRInfo &ri = *y355_convr_info(L.rid);
                kr.bn = ri.bn; kr.wn = ri.wn; kr.nt = ri.nt;
            }
            std::vector<char> pr(L.wr_bytes);
            y355_convg_pack(kr, nullptr, q_w, cout, cin, ksize, in_kbytes(h, o), L.cout_pad, pr.data());
            HIPCHK(hipMemcpy(L.wr_dev, pr.data(), pr.size(), hipMemcpyHostToDevice));
        }
    }
    if (h->front_graph && (L.op == 0 || L.op == 1)) {
        std::vector<int8_t> wf(16384);
        y355_pack_front(L.op == 0 ? q_w : nullptr, L.op == 1 ? q_w : nullptr, wf.data());
        if (L.op == 0) HIPCHK(hipMemcpy(h->wf_dev, wf.data(), 4096, hipMemcpyHostToDevice));
        else HIPCHK(hipMemcpy(h->wf_dev + 4096, wf.data() + 4096, 16384 - 4096, hipMemcpyHostToDevice));
        h->front_dirty = true;
    }
    L.q_b.assign(q_b, q_b + cout);
    L.e_w = e_w;
    L.e_b = e_b;
    {   // max over output channels of sum |q_w|: the layer's own bound on |acc| (127 * wabs) for the 32-bit-epilogue test
        long long wabs = 0;
        const size_t per = (size_t)cin * ksize * ksize;
        for (int c = 0; c < cout; ++c) {
            long long s = 0;
            for (size_t k = 0; k < per; ++k) s += std::abs((int)q_w[(size_t)c * per + k]);
            wabs = std::max(wabs, s);
        }
        L.wabs = wabs;
    }
    L.loaded = true;
    L.dirty = true;
    return 0;
}

// activation exponents of an int8 net: sa_in for the fp32 network input, sa[t] for tensor t (graph
// order of csrc/net.hip).  A max-pool output takes its input's exponent (the entry given for it is
// overridden); a concat buffer has ONE exponent that both producers requantise to.
extern "C" int y355_net_set_act_exponents(y355_net *h, int sa_in, const int32_t *sa, int n) {
    if (!h || !sa) return y355_fail(Y355_EINVAL, "null argument");
    if (h->bf) return y355_fail(Y355_EINVAL, "bf16 nets have no activation exponents");
    if (n != h->arch->ntensors) return y355_fail(Y355_EINVAL, "one exponent per tensor expected");
    for (int i = 0; i < n; ++i)
        if (sa[i] < -64 || sa[i] > 64) return y355_fail(Y355_EINVAL, "activation exponent out of range");
    if (sa_in < -64 || sa_in > 64) return y355_fail(Y355_EINVAL, "activation exponent out of range");
    h->sa_in = sa_in;
    h->sa.assign(sa, sa + n);
    for (int i = 0; i < h->arch->nops; ++i) {
        const OpDef &o = h->arch->ops[i];
        if (o.type == OP_POOL) h->sa[o.out] = h->sa[o.in];      // max-pool does not requantise
    }
    h->sa_ok = true;
    for (auto &L : h->L) L.dirty = true;
    return 0;
}

extern "C" int y355_net_get_act_exponents(y355_net *h, int32_t *sa_in, int32_t *sa, int n) {
    if (!h || !sa_in || !sa || n != h->arch->ntensors) return y355_fail(Y355_EINVAL, "bad argument");
    *sa_in = h->sa_in;
    for (int i = 0; i < n; ++i) sa[i] = h->sa[i];
    return 0;
}

static void act_fixed(int act, int *lk, int *neg_mul) {
    // LeakyReLU slope as neg_mul / 2^lk: 0.125 exactly; 0.1 ~ 205 / 2048 (build-defined)
    if (act == ACT_L125) { *lk = 3; *neg_mul = 1; }
    else if (act == ACT_L100) { *lk = 11; *neg_mul = 205; }
    else { *lk = 0; *neg_mul = 1; }
}

static int refresh_i8(y355_net *h) {
    if (!h->sa_ok) return y355_fail(Y355_ENOTREADY, "activation exponents not set (calibrate first)");
    for (int i = 0; i < h->arch->nops; ++i) {
        const OpDef &o = h->arch->ops[i];
        if (o.type != OP_CONV1 && o.type != OP_CONV) continue;
        NLayer &L = h->L[o.layer];
        if (!L.loaded) return y355_fail(Y355_ENOTREADY, "layer weights not loaded");
        if (!L.dirty) continue;
        const int sa_i = o.in < 0 ? h->sa_in : h->sa[o.in], sa_o = h->sa[o.out];
        const int F = std::max(sa_i + L.e_w, L.e_b);
        const int shl = F - sa_i - L.e_w, bshl = F - L.e_b;
        int lk, nm;
        act_fixed(o.act, &lk, &nm);
        const int sh = F + lk - sa_o;
        if (shl > 24 || bshl > 40 || sh > 62 || sh < -20) return y355_fail(Y355_ERANGE, "exponent gap too large for the fixed-point epilogue");
        // worst case |t'| (through the slope and a left shift / the rounding add) must stay below 2^62
        std::vector<long long> bw(L.cout_pad, 0);
        long double bmax = 0;
        for (int c = 0; c < L.cout; ++c) {
            bw[c] = (long long)L.q_b[c] * (1ll << bshl);
            bmax = std::max(bmax, (long double)std::llabs(bw[c]));
        }
        long double lim = ((long double)127 * 127 * o.ksize * o.ksize * o.cin) * std::ldexp(1.0L, shl) + bmax;
        lim *= std::max(std::ldexp(1.0L, lk), (long double)nm);
        lim = sh < 0 ? lim * std::ldexp(1.0L, -sh) : lim + std::ldexp(1.0L, sh);
        if (lim >= std::ldexp(1.0L, 62)) return y355_fail(Y355_ERANGE, "fixed-point epilogue exceeds 62 bits");
        L.rq.shl = shl;
        L.rq.sh = sh;
        L.rq.lk = lk;
        L.rq.neg_mul = nm;
        {
            // 32-bit epilogue (y355_requant_gen32) when |t| * max(2^max(0, lk - sh), neg_mul * 2^max(0, -sh)) + rounding < 2^31
            // |acc| <= 127 * (sum |q_w| of the channel): the layer's own weights give a tighter bound than 127 per weight
            const long double accmax = L.wabs > 0 ? (long double)127 * L.wabs : (long double)127 * 127 * o.ksize * o.ksize * o.cin;
            long double t32 = accmax * std::ldexp(1.0L, shl) + bmax;
            const long double fpos = std::ldexp(1.0L, std::max(0, lk - sh)), fneg = (long double)nm * std::ldexp(1.0L, std::max(0, -sh));
            t32 = t32 * std::max(fpos, fneg) + std::ldexp(1.0L, std::max(sh, 0));
            L.rq.narrow = (t32 < std::ldexp(1.0L, 31) && bmax < std::ldexp(1.0L, 31)) ? 1 : 0;
            L.rq.split = 0;
            if (!L.rq.narrow && sh >= 9 && sh <= 31 && nm >= 1 && nm < 4096) {
                // t itself and the positive branch fit 32 bits, only t * neg_mul does not: the negative branch goes in two halves
                const long double tb = accmax * std::ldexp(1.0L, shl) + bmax;
                const long double pos = tb * fpos + std::ldexp(1.0L, std::max(sh - lk, 0));
                const long double neg = (tb / 256 + 1) * nm + 256 + std::ldexp(1.0L, sh - 9);
                if (tb < std::ldexp(1.0L, 30) && pos < std::ldexp(1.0L, 31) && neg < std::ldexp(1.0L, 31) && bmax < std::ldexp(1.0L, 31))
                    L.rq.narrow = L.rq.split = 1;
            }
        }
        L.rq1 = Requant{};
        L.rq1.shl = shl;
        L.rq1.sh = sh;
        L.rq1.leaky = (lk || nm != 1) ? 1 : 0;
        L.rq1.lk = lk;
        L.rq1.neg_mul = nm;
        L.rq1.guard_log2 = 63;
        L.rq1.wide = 1;
        // the first layer runs conv1.hip's fast kernel when the general-slope epilogue fits 32 bits:
        //   |t| * max(2^max(0, lk - sh), neg_mul * 2^max(0, -sh)) + rounding < 2^31          (y355_requant_gen32)
        L.rq1.gen32 = 0;
        std::vector<int32_t> bt(L.cout_pad, 0);
        if (o.type == OP_CONV1) {
            long double t32 = ((long double)127 * 127 * o.ksize * o.ksize * o.cin) * std::ldexp(1.0L, shl) + bmax;
            const long double fpos = std::ldexp(1.0L, std::max(0, lk - sh)), fneg = (long double)nm * std::ldexp(1.0L, std::max(0, -sh));
            t32 = t32 * std::max(fpos, fneg) + std::ldexp(1.0L, std::max(sh, 0));
            if (t32 < std::ldexp(1.0L, 31) && bmax < std::ldexp(1.0L, 31)) {
                L.rq1.gen32 = 1;
                for (int c = 0; c < L.cout; ++c) bt[c] = (int32_t)bw[c];
            }
            // int8 nets have no float bias: the layer's bias_dev (4 bytes per channel) holds the 32-bit copy
            HIPCHK(hipMemcpyAsync(L.bias_dev, bt.data(), sizeof(int32_t) * L.cout_pad, hipMemcpyHostToDevice, h->stream));
        }
        if (h->front_graph && (i == 0 || i == 1)) {
            // the same epilogue for front.hip: t = acc * 2^shl + bias below 2^24 (exact in fp32) on the layer's own weights
            Requant &fr = i == 0 ? h->frq1 : h->frq2;
            fr = Requant{};
            fr.shl = shl; fr.sh = sh; fr.lk = lk; fr.neg_mul = nm; fr.leaky = (lk || nm != 1) ? 1 : 0; fr.guard_log2 = 63;
            const long double accmax = L.wabs > 0 ? (long double)127 * L.wabs : (long double)127 * 127 * o.ksize * o.ksize * o.cin;
            const long double tb = accmax * std::ldexp(1.0L, shl) + bmax;
            fr.tmax_log2 = 0;
            while (fr.tmax_log2 < 62 && std::ldexp(1.0L, fr.tmax_log2) <= tb) ++fr.tmax_log2;
            fr.negsafe = std::ldexp(tb * nm, -sh) <= 127.0L ? 1 : 0;
            // the magic-number rounding needs |t * slope| < 2^22 only where it does not saturate, and |sh| moderate
            fr.wide = (fr.tmax_log2 > 24 || bmax >= std::ldexp(1.0L, 24) || sh > 30 || sh - lk < -8 || nm < 1 || nm > (1 << lk)) ? 1 : 0;
            std::vector<int32_t> fb(i == 0 ? 16 : 32, 0);
            if (!fr.wide) for (int c = 0; c < L.cout; ++c) fb[c] = (int32_t)bw[c];
            HIPCHK(hipMemcpyAsync(i == 0 ? h->fb1_dev : h->fb2_dev, fb.data(), sizeof(int32_t) * fb.size(), hipMemcpyHostToDevice, h->stream));
            h->front_dirty = true;
        }
        HIPCHK(hipMemcpyAsync(L.bias_w_dev, bw.data(), sizeof(long long) * L.cout_pad, hipMemcpyHostToDevice, h->stream));
        HIPCHK(hipStreamSynchronize(h->stream));
        L.dirty = false;
    }
    if (h->front_graph && h->front_dirty) {
        h->front_ok = y355_front_eligible(h->frq1, h->frq2);
        h->front_dirty = false;
    }
    return 0;
}

// fp32 NCHW [B][3][H][W] -> bf16 NHWC16 with halo (channels 3..15 stay zero)
__global__ void input_bf16_kernel(const float *x, char *out, int B, int H, int W, int out_pb) {
    const size_t total = (size_t)B * H * W;
    const size_t plane = (size_t)H * W;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int xx = (int)(i % W), y = (int)((i / W) % H);
        const size_t b = i / plane;
        const float *src = x + b * 3 * plane + (size_t)y * W + xx;
        unsigned short h3[4] = {__builtin_bit_cast(unsigned short, (__bf16)src[0]), __builtin_bit_cast(unsigned short, (__bf16)src[plane]),
                                __builtin_bit_cast(unsigned short, (__bf16)src[2 * plane]), 0};
        uint2 u;
        u.x = (unsigned int)h3[0] | ((unsigned int)h3[1] << 16);
        u.y = (unsigned int)h3[2];
        *(uint2 *)(out + ((b * (H + 2) + y + 1) * (size_t)(W + 2) + xx + 1) * out_pb) = u;
    }
}
// utils.modules.reorg_layer (utils/modules.py:48-57) on bf16 NHWC: out[.., (sy*s+sx)*C + c] = in[s*y+sy][s*x+sx][c];
// 16 bytes (8 channels) per thread
__global__ void reorg_bf16_kernel(const char *in, char *out, int B, int Hin, int Win, int in_pb, int C, int out_pb, int out_off, int s) {
    const int Ho = Hin / s, Wo = Win / s, cg = C / 8;
    const size_t total = (size_t)B * Ho * Wo * s * s * cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const int k = (int)((i / cg) % (s * s));
        const int x = (int)((i / ((size_t)cg * s * s)) % Wo), y = (int)((i / ((size_t)cg * s * s * Wo)) % Ho);
        const size_t b = i / ((size_t)cg * s * s * Wo * Ho);
        const int sy = k / s, sx = k % s;
        const uint4 v = *(const uint4 *)(in + ((b * (Hin + 2) + (size_t)s * y + sy + 1) * (Win + 2) + (size_t)s * x + sx + 1) * in_pb + g * 16);
        *(uint4 *)(out + ((b * (Ho + 2) + y + 1) * (size_t)(Wo + 2) + x + 1) * out_pb + out_off + ((size_t)k * C + g * 8) * 2) = v;
    }
}

// utils.modules.SPP (utils/modules.py:66-72) on bf16 NHWC, in place in a 4C-channel buffer: channels [0, C) are x, the
// kernel writes max_pool 5 / 9 / 13 (stride 1, windows clipped to the map = -inf padding) to [C,2C), [2C,3C), [3C,4C).
// bf16 compares as fp32; 8 channels (16 bytes) per thread.
__global__ void spp_bf16_kernel(char *buf, int B, int H, int W, int pb, int C) {
    const int cg = C / 8;
    const size_t total = (size_t)B * H * W * cg;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int g = (int)(i % cg);
        const int x = (int)((i / cg) % W), y = (int)((i / ((size_t)cg * W)) % H);
        const size_t b = i / ((size_t)cg * W * H);
        float m5[8], m9[8], m13[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) m5[k] = m9[k] = m13[k] = -INFINITY;
        for (int dy = -6; dy <= 6; ++dy) {
            const int yy = y + dy;
            if (yy < 0 || yy >= H) continue;
            for (int dx = -6; dx <= 6; ++dx) {
                const int xx = x + dx;
                if (xx < 0 || xx >= W) continue;
                const uint4 v = *(const uint4 *)(buf + ((b * (H + 2) + yy + 1) * (size_t)(W + 2) + xx + 1) * pb + g * 16);
                const unsigned int u[4] = {v.x, v.y, v.z, v.w};
                const int r = max(abs(dy), abs(dx));
#pragma unroll
                for (int k = 0; k < 8; ++k) {
                    const float f = __uint_as_float((k & 1) ? (u[k >> 1] & 0xffff0000u) : (u[k >> 1] << 16));
                    m13[k] = fmaxf(m13[k], f);
                    if (r <= 4) m9[k] = fmaxf(m9[k], f);
                    if (r <= 2) m5[k] = fmaxf(m5[k], f);
                }
            }
        }
        char *o = buf + ((b * (H + 2) + y + 1) * (size_t)(W + 2) + x + 1) * pb + g * 16;
        auto pack = [](const float (&m)[8]) {
            uint4 r;
            unsigned int w[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) w[k] = (__float_as_uint(m[2 * k]) >> 16) | (__float_as_uint(m[2 * k + 1]) & 0xffff0000u);
            r.x = w[0]; r.y = w[1]; r.z = w[2]; r.w = w[3];
            return r;
        };
        *(uint4 *)(o + (size_t)C * 2) = pack(m5);
        *(uint4 *)(o + (size_t)C * 4) = pack(m9);
        *(uint4 *)(o + (size_t)C * 6) = pack(m13);
    }
}

// bytes of input channels a convolution consumes per pixel (its K extent): o.cin rounded up to the channel quantum --
// the whole pixel for ordinary tensors, a leading channel range for concat buffers read before they are complete
static int in_kbytes(const y355_net *h, const OpDef &o) {
    const int cq = h->bf ? 16 : 32;
    int c = (o.cin + cq - 1) / cq * cq;
    if (h->bf && c > 16) c = (c + 31) / 32 * 32;          // 64-byte k-steps beyond the thin (16-channel) path
    return c * h->es;
}

static float act_slope(int act) { return act == ACT_L125 ? 0.125f : act == ACT_L100 ? 0.1f : 1.0f; }

static int run_op(y355_net *h, int i, int B, const float *x_dev) {
    const OpDef &o = h->arch->ops[i];
    hipStream_t s = h->stream;
    if (o.type == OP_CONV1) {
        const NLayer &L = h->L[o.layer];
        if (!L.loaded) return y355_fail(Y355_ENOTREADY, "layer weights not loaded");
        if (!h->bf) {
            Conv1Params p{};
            p.x = x_dev;
            p.out = (int8_t *)h->T[o.out].dev;
            p.out_pb = (int)h->T[o.out].pb;
            p.w = (const int8_t *)h->w0_dev;
            p.bias_w = L.bias_w_dev;
            p.bias_t = L.rq1.gen32 ? (const int *)L.bias_dev : nullptr;
            p.ctr = h->ctr_dev + i;
            p.B = B;
            p.H = h->cfg.height;
            p.W = h->cfg.width;
            y355_conv1_tiles(p.H, p.W, &p.tiles_x, &p.tiles_y);
            p.in_scale = std::ldexp(1.0f, h->sa_in);
            p.rq = L.rq1;
            p.mode = 0;
            p.guard = 0;
            y355_launch_conv1(p, s);
            HIPCHK(hipGetLastError());
            return 0;
        }
        Conv1FParams p{};
        p.x = x_dev;
        p.out = h->T[o.out].dev;
        p.w = h->w0_dev;
        p.bias = L.bias_dev;
        p.B = B;
        p.H = h->cfg.height;
        p.W = h->cfg.width;
        y355_conv1f_tiles(p.H, p.W, &p.tiles_x, &p.tiles_y);
        p.slope = act_slope(o.act);
        y355_launch_conv1f(p, s);
    } else if (o.type == OP_CONV) {
        const NLayer &L = h->L[o.layer];
        if (!L.loaded) return y355_fail(Y355_ENOTREADY, "layer weights not loaded");
        const Tensor &ti = h->T[o.in], &to = h->T[o.out];
        const ConvGInfo &ki = *y355_convg_kernel(h->bf, L.kid);
        ConvGParams p{};
        p.in = ti.dev;
        p.out = to.dev;
        p.w = L.w_dev;
        p.bias_f = L.bias_dev;
        p.bias_w = L.bias_w_dev;
        p.ctr = h->ctr_dev + i;
        p.rq = L.rq;
        p.B = B;
        p.H = ti.H;
        p.W = ti.W;
        p.in_pb = (int)ti.pb;
        p.nchunks = in_kbytes(h, o) / ki.chb;
        p.out_pb = (int)to.pb;
        p.out_off = o.choff * ((to.pred && h->bf) ? 4 : h->es);
        p.out_halo = to.halo;
        const int Ho = o.stride2 ? (ti.H + 1) / 2 : ti.H, Wo = o.stride2 ? (ti.W + 1) / 2 : ti.W;
        p.tiles_x = (Wo + ki.tw - 1) / ki.tw;
        p.tiles_y = (Ho + ki.th - 1) / ki.th;
        if (o.res1) {
            const Tensor &tr = h->T[o.res1 - 1];
            p.res = tr.dev;
            p.res_pb = (int)tr.pb;
            p.res_off = 0;
        }
        p.nblk = L.cout_pad / ki.bn;
        p.taps = o.ksize * o.ksize;
        p.slope = act_slope(o.act);
        p.out_f32 = to.pred && h->bf;
        p.grid_limit = h->tput_wgs;
        if (L.rid == -2) {                                     // int8 1x1: pointwise kernel (convr.hip)
            ConvGParams q = p;
            q.w = L.wr_dev;
            q.nblk = L.cout_pad / 64;
            if (y355_launch_pw_i8(q, s)) {
                HIPCHK(hipGetLastError());
                return 0;
            }
        }
        if (L.pbid >= 0 && !h->no_pxb) {                        // bf16 thin 3x3: weights in registers, pixels as the B operand (convpxb.hip)
            ConvGParams q = p;
            q.w = L.wpb_dev;
            if (y355_launch_convpxb(L.pbid, q, s)) {
                HIPCHK(hipGetLastError());
                return 0;
            }
        }
        if (L.rid >= 0) {                                      // 3x3: weights through an LDS ring (convr.hip)
            ConvGParams q = p;
            q.w = L.wr_dev;
            q.nblk = L.cout_pad / y355_convr_info(L.rid)->bn;
            if (y355_launch_convr(L.rid, q, h->cfg.device_id, s)) {
                HIPCHK(hipGetLastError());
                return 0;
            }
        }
        ki.launch(p, p.tiles_x * p.tiles_y * p.nblk * B, s);
    } else if (o.type == OP_POOL) {
        const Tensor &ti = h->T[o.in], &to = h->T[o.out];
        const int stride = o.pool ? 1 : 2;
        const size_t total = (size_t)B * to.H * to.W * (o.cin * h->es / 16);
        const int blocks = (int)std::min<size_t>((total + 255) / 256, 4096);
        if (h->bf)
            hipLaunchKernelGGL(pool_bf16_kernel, dim3(blocks), dim3(256), 0, s, ti.dev, to.dev, B, ti.H, ti.W, (int)ti.pb,
                               o.cin * h->es, to.H, to.W, (int)to.pb, stride);
        else
            hipLaunchKernelGGL(pool_i8_kernel, dim3(blocks), dim3(256), 0, s, ti.dev, to.dev, B, ti.H, ti.W, (int)ti.pb,
                               o.cin * h->es, to.H, to.W, (int)to.pb, stride);
    } else if (o.type == OP_INPUT) {
        const Tensor &to = h->T[o.out];
        const size_t total = (size_t)B * to.H * to.W;
        hipLaunchKernelGGL(input_bf16_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0, s, x_dev, to.dev, B,
                           to.H, to.W, (int)to.pb);
    } else if (o.type == OP_SPP) {
        const Tensor &t = h->T[o.in];
        const size_t total = (size_t)B * t.H * t.W * (o.cin / 8);
        hipLaunchKernelGGL(spp_bf16_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0, s, t.dev, B, t.H, t.W,
                           (int)t.pb, o.cin);
    } else if (o.type == OP_REORG) {
        const Tensor &ti = h->T[o.in], &to = h->T[o.out];
        const size_t total = (size_t)B * to.H * to.W * o.ksize * o.ksize * (o.cin / 8);
        hipLaunchKernelGGL(reorg_bf16_kernel, dim3((int)std::min<size_t>((total + 255) / 256, 8192)), dim3(256), 0, s, ti.dev, to.dev, B,
                           ti.H, ti.W, (int)ti.pb, o.cin, (int)to.pb, o.choff * h->es, o.ksize);
    } else {
        const Tensor &ti = h->T[o.in], &to = h->T[o.out];
        const size_t total = (size_t)B * to.H * to.W * o.cin;
        const int blocks = (int)std::min<size_t>((total + 255) / 256, 8192);
        const float ry = (float)(ti.H - 1) / (float)(to.H - 1), rx = (float)(ti.W - 1) / (float)(to.W - 1);
        if (h->bf)
            hipLaunchKernelGGL(upsample_bf16_kernel, dim3(blocks), dim3(256), 0, s, ti.dev, to.dev, B, ti.H, ti.W, (int)ti.pb, o.cin,
                               (int)to.pb, o.choff * h->es, ry, rx);
        else if (o.cin % 16 == 0 && ti.pb % 16 == 0 && to.pb % 16 == 0 && (o.choff * h->es) % 16 == 0 &&
                 (long long)B * to.H * to.W * (o.cin / 16) < (1ll << 31)) {
            const int items = B * to.H * to.W * (o.cin / 16);
            hipLaunchKernelGGL(upsample_i8x16_kernel, dim3((items + 255) / 256), dim3(256), 0, s, ti.dev, to.dev, B, ti.H, ti.W, (int)ti.pb,
                               o.cin, (int)to.pb, o.choff * h->es, ry, rx, std::ldexp(1.0f, h->sa[o.out] - h->sa[o.in]));
        } else
            hipLaunchKernelGGL(upsample_i8_kernel, dim3(blocks), dim3(256), 0, s, ti.dev, to.dev, B, ti.H, ti.W, (int)ti.pb, o.cin,
                               (int)to.pb, o.choff * h->es, ry, rx, std::ldexp(1.0f, h->sa[o.out] - h->sa[o.in]));
    }
    HIPCHK(hipGetLastError());
    return 0;
}

static HeadParams net_head_params(y355_net *h, float *ob, float *os, int *oc, int *on) {
    HeadParams p{};
    const ArchDef &A = *h->arch;
    p.nlev = A.nlev;
    for (int l = 0; l < A.nlev; ++l) {
        const Tensor &t = h->T[A.pred_t[l]];
        HeadLevel &lv = p.lev[l];
        lv.pred = h->bf ? nullptr : (const int8_t *)t.dev;
        lv.pred_f = h->bf ? (const float *)t.dev : nullptr;
        lv.cstride = t.Cpad;
        lv.Hs = t.H;
        lv.Ws = t.W;
        lv.stride = A.stride[l];
        lv.dq = h->bf ? 1.0f : std::ldexp(1.0f, -h->sa[A.pred_t[l]]);
        for (int i = 0; i < 2 * h->cfg.num_anchors; ++i) lv.anchors[i] = h->cfg.anchors[l * 2 * h->cfg.num_anchors + i];
    }
    p.A = h->cfg.num_anchors;
    p.C = h->cfg.num_classes;
    // slim-YOLOv2 anchors are in grid units (models/slim_yolo_v2.py:126), tiny-v3's in pixels (tiny_yolo_v3.py:85)
    p.wh_mul = h->cfg.arch == Y355_ARCH_SLIM_V2 ? 16.0f : (h->cfg.arch == Y355_ARCH_YOLO_V2 ? 32.0f : 1.0f);
    // fp32 / multi-level heads: box sizes spread over many octaves per anchor -> group by area
    // (YOLOv3tiny int8, B = 128: NMS 1.08 -> 0.50 ms; SlimYOLOv2 bf16: 0.45 -> 0.32 ms)
    p.group_by_area = 1;
    p.pairs_wgs = h->tput_wgs > 0 ? 1 : 0;        // throughput mode: the pair walk holds one CU per image
    p.Hb = std::min(16, h->T[A.pred_t[0]].H);
    p.Wb = std::min(16, h->T[A.pred_t[0]].W);
    p.in_w = (float)h->cfg.width;
    p.in_h = (float)h->cfg.height;
    p.conf_thresh = h->cfg.conf_thresh;
    p.nms_thresh = h->cfg.nms_thresh;
    p.cand_box = h->cand_box;
    p.cand_score = h->cand_score;
    p.cand_cls = h->cand_cls;
    p.max_det = h->max_det;
    p.out_box = ob;
    p.out_score = os;
    p.out_cls = oc;
    p.out_count = on;
    return p;
}

extern "C" int y355_net_forward(y355_net *h, const float *x_dev, int batch, int flags, float *boxes_dev, float *scores_dev,
                                int32_t *cls_dev, int32_t *count_dev) {
    if (!h || !x_dev || !boxes_dev || !scores_dev || !cls_dev || !count_dev) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 1 || batch > h->cfg.max_batch) return y355_fail(Y355_EINVAL, "batch out of range");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    const bool prof = h->profile != 0;
    const int nops = h->arch->nops;
    if (!h->bf) {
        if (int rc = refresh_i8(h)) return rc;
        bool need_zero = false;
        h->ctr_dev = h->ctrs.begin(&need_zero);        // steady state: zeroed by the previous forward's fused front end
        if (need_zero) HIPCHK((hipError_t)y355_zero_counters(h->ctr_dev, nops + 1, h->stream));
    }
    // tap forwards (parity tests read every tensor) run the first two layers one by one: the fused launch does not write conv1's map
    const bool fuse_front = h->front_graph && (h->bf || h->front_ok) && !(flags & Y355_F_TAP);
    h->t0_skipped = fuse_front;
    for (int i = 0; i < nops; ++i) {
        if (prof) HIPCHK(hipEventRecord(h->ev[i], h->stream));
        if (fuse_front && i < 2) {
            if (i == 0) {
                if (!h->L[h->arch->ops[0].layer].loaded || !h->L[h->arch->ops[1].layer].loaded)
                    return y355_fail(Y355_ENOTREADY, "layer weights not loaded");
                if (h->bf) {
                    const OpDef &o0 = h->arch->ops[0], &o1 = h->arch->ops[1];
                    FrontBParams fb{};
                    fb.x = x_dev;
                    fb.out = h->T[o1.out].dev;
                    fb.out_pb = (int)h->T[o1.out].pb;
                    fb.wf = (const char *)h->wf_dev;
                    fb.bias1 = h->L[o0.layer].bias_dev;
                    fb.bias2 = h->L[o1.layer].bias_dev;
                    fb.B = batch;
                    fb.H = h->cfg.height;
                    fb.W = h->cfg.width;
                    y355_frontb_tiles(fb.H, fb.W, &fb.tiles_x, &fb.tiles_y);
                    fb.slope1 = act_slope(o0.act);
                    fb.slope2 = act_slope(o1.act);
                    y355_launch_frontb(fb, h->stream);
                    HIPCHK(hipGetLastError());
                    continue;
                }
                FrontParams fp{};
                fp.x = x_dev;
                fp.out = (int8_t *)h->T[h->arch->ops[1].out].dev;
                fp.wf = h->wf_dev;
                fp.bias1 = h->fb1_dev;
                fp.bias2 = h->fb2_dev;
                fp.ctr = h->ctr_dev;                                // [0] conv1 (+ input), [1] conv2: the two ops' own counters
                fp.zero_next = (unsigned long long *)h->ctrs.other_zeroed_by_front();
                fp.zero_n = (nops + 1) * (int)(sizeof(Counters) / 8);
                fp.B = batch;
                fp.H = h->cfg.height;
                fp.W = h->cfg.width;
                y355_front_tiles(fp.H, fp.W, &fp.tiles_x, &fp.tiles_y);
                fp.in_scale = std::ldexp(1.0f, h->sa_in);
                fp.rq1 = h->frq1;
                fp.rq2 = h->frq2;
                y355_launch_front(fp, h->stream);
                HIPCHK(hipGetLastError());
            }
            continue;
        }
        if (int rc = run_op(h, i, batch, x_dev)) return rc;
    }
    if (prof) HIPCHK(hipEventRecord(h->ev[nops], h->stream));
    HeadParams hp = net_head_params(h, boxes_dev, scores_dev, cls_dev, count_dev);
    if (!(flags & Y355_F_TAP)) { hp.cand_box = nullptr; hp.cand_score = nullptr; hp.cand_cls = nullptr; }
    y355_launch_head_nms(hp, batch, h->ws, h->stream, prof ? h->ev[nops + 1] : nullptr);
    HIPCHK(hipGetLastError());
    if (prof) HIPCHK(hipEventRecord(h->ev[nops + 2], h->stream));
    return 0;
}

extern "C" int y355_net_get_candidates(y355_net *h, int batch, float *boxes, float *scores, int32_t *cls) {
    if (!h || !boxes || !scores || !cls) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 1 || batch > h->cfg.max_batch) return y355_fail(Y355_EINVAL, "batch out of range");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(boxes, h->cand_box, sizeof(float) * 4 * h->N * batch, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(scores, h->cand_score, sizeof(float) * h->N * batch, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cls, h->cand_cls, sizeof(int) * h->N * batch, hipMemcpyDeviceToHost));
    return 0;
}

// parity tap: tensor idx as fp32 NCHW [B][C][H][W] on the host (halo and channel padding stripped)
extern "C" int y355_net_get_tensor(y355_net *h, int idx, int batch, float *dst) {
    if (!h || !dst || idx < 0 || idx >= h->arch->ntensors) return y355_fail(Y355_EINVAL, "bad argument");
    if (batch < 1 || batch > h->cfg.max_batch) return y355_fail(Y355_EINVAL, "batch out of range");
    if (idx == h->arch->ops[0].out && h->t0_skipped)
        return y355_fail(Y355_ENOTREADY, "the first layer's map of the last forward was not written (fused front end): "
                                         "run the forward with Y355_F_TAP to read it");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    const Tensor &t = h->T[idx];
    const int Hp = t.H + 2 * t.halo, Wp = t.W + 2 * t.halo;
    std::vector<char> tmp((size_t)batch * Hp * Wp * t.pb);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(tmp.data(), t.dev, tmp.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < batch; ++b)
        for (int c = 0; c < t.C; ++c)
            for (int y = 0; y < t.H; ++y)
                for (int x = 0; x < t.W; ++x) {
                    const char *src = tmp.data() + (((size_t)b * Hp + y + t.halo) * Wp + x + t.halo) * t.pb;
                    float v;
                    if (!h->bf) {
                        v = std::ldexp((float)*(const signed char *)(src + c), -h->sa[idx]);
                    } else if (t.pred) {
                        memcpy(&v, src + (size_t)c * 4, 4);
                    } else {
                        unsigned short hbits;
                        memcpy(&hbits, src + (size_t)c * 2, 2);
                        const unsigned int u = (unsigned int)hbits << 16;
                        memcpy(&v, &u, 4);
                    }
                    dst[(((size_t)b * t.C + c) * t.H + y) * t.W + x] = v;
                }
    return 0;
}

// max |value| of tensor idx (calibration tap for the int8 recipe); synchronous
extern "C" int y355_net_tensor_absmax(y355_net *h, int idx, int batch, float *out_max) {
    if (!h || !out_max || idx < 0 || idx >= h->arch->ntensors) return y355_fail(Y355_EINVAL, "bad argument");
    if (batch < 1 || batch > h->cfg.max_batch) return y355_fail(Y355_EINVAL, "batch out of range");
    const Tensor &t = h->T[idx];
    if (!h->bf) return y355_fail(Y355_EINVAL, "absmax taps exist on bf16 nets (the calibration run)");
    if (t.pred) return y355_fail(Y355_EINVAL, "prediction maps are fp32: read them with y355_net_get_tensor");
    if (idx == h->arch->ops[0].out && h->t0_skipped)
        return y355_fail(Y355_ENOTREADY, "the first layer's map of the last forward was not written (fused front end): "
                                         "calibrate from a forward with Y355_F_TAP");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipMemsetAsync(h->absmax_dev, 0, 16, h->stream));
    const size_t n = (size_t)batch * (t.H + 2) * (t.W + 2) * t.Cpad;
    hipLaunchKernelGGL(absmax_bf16_kernel, dim3(1024), dim3(256), 0, h->stream, t.dev, n, h->absmax_dev);
    HIPCHK(hipGetLastError());
    unsigned int bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, h->absmax_dev, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(out_max, &bits, 4);
    return 0;
}

// values clamped to +-127 in the last forward of an int8 net (input quantisation included); synchronous
extern "C" int y355_net_counters(y355_net *h, int64_t *saturated) {
    if (!h || !saturated) return y355_fail(Y355_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    std::vector<Counters> c(h->arch->nops + 1);
    HIPCHK(hipMemcpyAsync(c.data(), h->ctr_dev, sizeof(Counters) * c.size(), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    int64_t s = 0;
    for (auto &k : c) s += (int64_t)k.sat + (int64_t)k.in_sat;
    *saturated = s;
    return 0;
}

// heads with more than 4096 anchors per image: 1 if, in any forward since the last call, more than 4096 anchors of an
// image passed conf_thresh (the excess was dropped); synchronous; clears the flags
extern "C" int y355_net_overflow(y355_net *h, int *overflow) {
    if (!h || !overflow) return y355_fail(Y355_EINVAL, "null argument");
    *overflow = 0;
    if (!h->ws.ovf) return 0;
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    std::vector<int> v(h->cfg.max_batch, 0);
    HIPCHK(hipMemcpy(v.data(), h->ws.ovf, sizeof(int) * v.size(), hipMemcpyDeviceToHost));
    for (int x : v) *overflow |= x != 0;
    if (*overflow) HIPCHK(hipMemset(h->ws.ovf, 0, sizeof(int) * v.size()));
    return 0;
}

// diagnostics: candidates and suppressing pairs per image of the last forward's NMS (count[batch], nedges[2 * batch]: list
// length, overflow / abort flag); synchronous
extern "C" int y355_net_debug_nms(y355_net *h, int batch, int32_t *count, int32_t *nedges) {
    if (!h || !count || !nedges || batch < 1 || batch > h->cfg.max_batch) return y355_fail(Y355_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(count, h->ws.count, sizeof(int) * batch, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(nedges, h->ws.nedges, sizeof(int) * 2 * batch, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int y355_net_sync(y355_net *h) {
    if (!h) return y355_fail(Y355_EINVAL, "null net");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int y355_net_profile(y355_net *h, int enable) {
    if (!h) return y355_fail(Y355_EINVAL, "null net");
    h->profile = enable;
    return 0;
}

extern "C" int y355_net_num_timers(y355_net *h) { return h ? h->arch->nops + 2 : Y355_EINVAL; }

extern "C" int y355_net_profile_get(y355_net *h, float *ms) {
    if (!h || !ms) return y355_fail(Y355_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    const int n = h->arch->nops + 2;
    HIPCHK(hipEventSynchronize(h->ev[n]));
    for (int i = 0; i < n; ++i) HIPCHK(hipEventElapsedTime(&ms[i], h->ev[i], h->ev[i + 1]));
    return 0;
}

// ------------------------------------------------------------------------------------------
// Stand-alone detection head on fp32 prediction maps (operator API of the wider model families):
// models/yolo_v2.py:183-210 (one level, anchors in grid units: wh_mul = stride), models/tiny_yolo_v3.py /
// models/yolo_v3.py:207-262 (anchors in pixels: wh_mul = 1): decode, sigmoid(obj) * softmax(cls), threshold,
// per-class NMS, survivors in anchor-index order.  Host pointers, synchronous; pred[l] is NCHW
// [B][A*(5+C)][hs[l]][ws[l]] like the reference's tensors.
extern "C" int y355_head_f32(int device_id, int nlev, const float *const *pred, const int *hs, const int *ws, const float *strides,
                             const float *anchors, int num_anchors, int num_classes, int in_h, int in_w, float wh_mul,
                             float conf_thresh, float nms_thresh, int batch, int max_det, float *boxes, float *scores,
                             int32_t *cls, int32_t *count) {
    if (!pred || !hs || !ws || !strides || !anchors || !boxes || !scores || !cls || !count) return y355_fail(Y355_EINVAL, "null argument");
    if (nlev < 1 || nlev > 3) return y355_fail(Y355_EINVAL, "1 to 3 prediction levels");
    if (num_anchors < 1 || num_anchors * nlev > Y355_HEAD_MAXA || num_classes < 1 || batch < 1 || max_det < 1)
        return y355_fail(Y355_EINVAL, "bad anchors / classes / batch / max_det");
    const int predc = num_anchors * (5 + num_classes);
    if (predc > 256) return y355_fail(Y355_EINVAL, "A*(5+C) > 256 not supported");
    int N = 0;
    for (int l = 0; l < nlev; ++l) {
        if (!pred[l] || hs[l] < 1 || ws[l] < 1) return y355_fail(Y355_EINVAL, "bad prediction level");
        N += hs[l] * ws[l] * num_anchors;
    }
    if (N > 16 * Y355_NMS_CAP) return y355_fail(Y355_EINVAL, "more than 65536 anchors per image not supported");
    const bool large = N > Y355_NMS_CAP;          // threshold-then-compact in front of the sort (head_nms.hip)
    if (max_det > (large ? Y355_NMS_CAP : N)) max_det = large ? Y355_NMS_CAP : N;
    HIPCHK(hipSetDevice(device_id));
    if (int e = y355_prepare_kernels()) return e;
    std::vector<void *> bufs;
    auto alloc = [&](void **p, size_t bytes, bool zero) -> int {
        if (hipMalloc(p, bytes ? bytes : 16) != hipSuccess) return 1;
        bufs.push_back(*p);
        if (zero && hipMemset(*p, 0, bytes ? bytes : 16) != hipSuccess) return 1;
        return 0;
    };
    auto release = [&]() { for (void *q : bufs) (void)hipFree(q); };
    const int B = batch;
    const size_t cap = Y355_NMS_CAP;
    const int cpad = predc <= 64 ? 64 : predc <= 128 ? 128 : 256;
    y355_head_ws w{};
    int rc = 0;
    float *d_pred[3] = {nullptr, nullptr, nullptr};
    for (int l = 0; l < nlev && !rc; ++l) {
        // NCHW -> NHWC with cpad channels
        const size_t px = (size_t)hs[l] * ws[l];
        std::vector<float> t((size_t)B * px * cpad, 0.f);
        for (int b = 0; b < B; ++b)
            for (int c = 0; c < predc; ++c) {
                const float *src = pred[l] + ((size_t)b * predc + c) * px;
                for (size_t i = 0; i < px; ++i) t[((size_t)b * px + i) * cpad + c] = src[i];
            }
        rc = alloc((void **)&d_pred[l], t.size() * 4 + 1024, false);
        if (!rc && hipMemcpy(d_pred[l], t.data(), t.size() * 4, hipMemcpyHostToDevice) != hipSuccess) rc = 1;
    }
    float *d_box = nullptr, *d_score = nullptr;
    int *d_cls = nullptr, *d_count = nullptr;
    if (!rc) rc = alloc(&w.cbox, sizeof(float) * 4 * cap * B, false);
    if (!rc) rc = alloc(&w.cscore, sizeof(float) * cap * B, false);
    if (!rc) rc = alloc(&w.ccls, sizeof(int) * cap * B, false);
    if (!rc) rc = alloc(&w.corig, sizeof(int) * cap * B, false);
    if (!rc) rc = alloc(&w.count, sizeof(int) * B, true);
    if (!rc) rc = alloc(&w.edges, sizeof(unsigned int) * (size_t)Y355_HEAD_EDGE_CAP * B, false);
    if (!rc) rc = alloc(&w.nedges, sizeof(int) * 2 * (size_t)B, true);
    if (!rc) rc = alloc(&w.binstart, sizeof(int) * (cap + 8) * B, true);
    if (!rc) rc = alloc(&w.astat, sizeof(float) * 4 * Y355_HEAD_MAXG * B, true);
    if (!rc) rc = alloc(&w.tiny, sizeof(int) * cap * B, true);
    if (!rc) rc = alloc(&w.ntiny, sizeof(int) * B, true);
    if (!rc) rc = alloc(&w.dbox, sizeof(float) * 4 * cap * B, true);
    if (!rc) rc = alloc(&w.dscore, sizeof(float) * cap * B, true);
    if (!rc) rc = alloc(&w.dcls, sizeof(int) * cap * B, true);
    if (!rc) rc = alloc(&w.ctype, sizeof(int) * cap * B, true);
    if (large) {
        w.rstride = (N + 3) / 4 * 4;
        if (!rc) rc = alloc(&w.rbox, sizeof(float) * 4 * (size_t)w.rstride * B, true);
        if (!rc) rc = alloc(&w.rscore, sizeof(float) * (size_t)w.rstride * B, true);
        if (!rc) rc = alloc(&w.rcls, sizeof(int) * (size_t)w.rstride * B, true);
        if (!rc) rc = alloc(&w.rcount, sizeof(int) * B, true);
        if (!rc) rc = alloc(&w.ovf, sizeof(int) * B, true);
    }
    if (!rc) rc = alloc((void **)&d_box, sizeof(float) * 4 * (size_t)max_det * B, true);
    if (!rc) rc = alloc((void **)&d_score, sizeof(float) * (size_t)max_det * B, true);
    if (!rc) rc = alloc((void **)&d_cls, sizeof(int) * (size_t)max_det * B, true);
    if (!rc) rc = alloc((void **)&d_count, sizeof(int) * B, true);
    if (rc) { release(); return y355_fail(Y355_EHIP, "device allocation failed"); }
    HeadParams p{};
    p.nlev = nlev;
    for (int l = 0; l < nlev; ++l) {
        HeadLevel &lv = p.lev[l];
        lv.pred = nullptr;
        lv.pred_f = d_pred[l];
        lv.cstride = cpad;
        lv.Hs = hs[l];
        lv.Ws = ws[l];
        lv.stride = strides[l];
        lv.dq = 1.0f;
        for (int i = 0; i < 2 * num_anchors; ++i) lv.anchors[i] = anchors[l * 2 * num_anchors + i];
    }
    p.A = num_anchors;
    p.C = num_classes;
    p.wh_mul = wh_mul;
    p.group_by_area = 1;
    p.Hb = std::min(16, hs[0]);
    p.Wb = std::min(16, ws[0]);
    p.in_w = (float)in_w;
    p.in_h = (float)in_h;
    p.conf_thresh = conf_thresh;
    p.nms_thresh = nms_thresh;
    p.max_det = max_det;
    p.out_box = d_box;
    p.out_score = d_score;
    p.out_cls = d_cls;
    p.out_count = d_count;
    y355_launch_head_nms(p, B, w, 0, nullptr);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess) e = hipMemcpy(boxes, d_box, sizeof(float) * 4 * (size_t)max_det * B, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(scores, d_score, sizeof(float) * (size_t)max_det * B, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(cls, d_cls, sizeof(int) * (size_t)max_det * B, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(count, d_count, sizeof(int) * B, hipMemcpyDeviceToHost);
    bool overflow = false;
    if (e == hipSuccess && large) {
        std::vector<int> ovf(B, 0);
        e = hipMemcpy(ovf.data(), w.ovf, sizeof(int) * B, hipMemcpyDeviceToHost);
        for (int v : ovf) overflow |= v != 0;
    }
    release();
    if (overflow) return y355_fail(Y355_EINVAL, "more than 4096 anchors of an image pass conf_thresh: raise the threshold");
    if (e != hipSuccess) return y355_fail(Y355_EHIP, std::string("head: ") + hipGetErrorString(e));
    return 0;
}

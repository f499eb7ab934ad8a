// yolo355 -- shared device/host declarations (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <hip/hip_ext.h>

typedef int v4i __attribute__((ext_vector_type(4)));
#define Y355_STAMP_ROWS 4096      // rows of 32 stamps in the diagnostic builds' stamp buffer (y355_debug_stamps)

// launch; with both events given the launch records its own start / end timestamps into them (profile mode 2: the
// duration rocprofv3 reports for the kernel, without the gap to the neighbouring launches)
#define Y355_LAUNCH(kernel, grid, block, lds, stream, e0, e1, ...)                                                       \
    do {                                                                                                                 \
        if ((e0) && (e1)) hipExtLaunchKernelGGL(kernel, grid, block, lds, stream, (hipEvent_t)(e0), (hipEvent_t)(e1), 0, __VA_ARGS__); \
        else hipLaunchKernelGGL(kernel, grid, block, lds, stream, __VA_ARGS__);                                          \
    } while (0)

// Fixed-point epilogue of one fused layer (DESIGN.md "requantisation"):
//   t  = (acc << shl) + bias_t[c]        bias_t = q_b << (F - e_b),  F = max(sa_in + e_w, e_b)
//   t' = leaky ? max(t, 8 t) : t         LeakyReLU(0.125) scaled by 8, F' = F + 3
//   q  = clamp(RNE(t' * 2^-sh), +-127)   sh = F' - sa_out
// Restates models/slim_yolo_v2.py:220-231 + :33-38 exactly in integers;
// shift composition as programmed by c_embedding/yolo_forward.c:233-257.
struct Requant {
    int shl;
    int sh;
    int leaky;
    int guard_log2;   // guard trips when |t'| >= 2^guard_log2 ( >= 63: never )
    int wide;         // 1: worst-case |t'| does not fit 30 bits -> 64-bit epilogue kernels
    // branch-free form of the same pipeline (32-bit path), filled by the host:
    //   t' = max(t, t << lk)                       lk = leaky ? 3 : 0
    //   q  = ((t' << sh_l) + hm1 + bfe(t', sh_r, bw)) >> sh_r
    // with sh_r = max(sh,0), sh_l = max(-sh,0), hm1 = sh>0 ? 2^(sh-1)-1 : 0, bw = sh>0 ? 1 : 0
    int lk, sh_l, sh_r, hm1, bw;
    // general LeakyReLU slope neg_mul / 2^lk (1 / 2^3 = the reference's 0.125); only the 64-bit
    // epilogues (y355_pre) honour neg_mul != 1
    int neg_mul;
    // 1: the general slope fits 32 bits (host-checked: |t| * max(2^max(0, lk - sh), neg_mul * 2^max(0, -sh)) < 2^31):
    //   q = t >= 0 ? rne(t * 2^(lk - sh)) : rne(t * neg_mul * 2^-sh)          (y355_requant_gen32; first layer of y355_net)
    int gen32;
    // |t| = |(acc << shl) + bias_t| < 2^tmax_log2 for worst-case operands (<= 24: t is exact in fp32, front.hip's epilogue)
    int tmax_log2;
    // gen32 only: 1 = the negative branch's product t * neg_mul does not fit 32 bits but t does; it is taken in two halves
    // (y355_requant_gen32): needs sh >= 9 and (|t| / 256 + 1) * neg_mul + 256 < 2^31 (host-checked)
    int split;
    // 1: the LeakyReLU's negative branch cannot leave [-127, 127]: |t| * neg_mul * 2^-sh <= 127 for the worst-case |t| of these weights
    // (host-checked on the exact bound, not on 2^tmax_log2): front.hip's hot passes then do not track that branch's minimum
    int negsafe;
};

__device__ __forceinline__ int y355_rne_shift32(int x, int s) {            // s wave-uniform
    if (s <= 0) return x << (-s);
    return (x + (1 << (s - 1)) - 1 + ((x >> s) & 1)) >> s;
}
// q before clamping for a LeakyReLU slope neg_mul / 2^lk that is not a power of two, 32-bit (Requant::gen32)
__device__ __forceinline__ int y355_requant_gen32(int acc, int bias, const Requant &rq) {
    const int t = (acc << rq.shl) + bias;
    const int qp = y355_rne_shift32(t, rq.sh - rq.lk);
    int qn;
    if (rq.split) {                                    // wave-uniform
        // t = 256 hi + lo (0 <= lo < 256):  t * neg_mul = 256 A + rem  with  A = hi * neg_mul + (lo * neg_mul >> 8),
        // rem = lo * neg_mul & 255.  RNE((256 A + rem) / 2^sh) = RNE of A / 2^(sh - 8) in which a non-zero rem turns an exact
        // tie into "above the tie": the usual half - 1 + lsb rounding add with lsb forced to 1
        const int hi = t >> 8, m2 = (t & 255) * rq.neg_mul;
        const int A = hi * rq.neg_mul + (m2 >> 8);
        const int s = rq.sh - 8;
        qn = (A + ((1 << (s - 1)) - 1) + ((((m2 & 255) + 255) >> 8) | ((A >> s) & 1))) >> s;
    } else {
        qn = y355_rne_shift32(t * rq.neg_mul, rq.sh);
    }
    return t >= 0 ? qp : qn;
}

// q before clamping, 32-bit, no branches (production kernels)
__device__ __forceinline__ int y355_requant_fast(int acc, int bias, const Requant &rq) {
    int t = (acc << rq.shl) + bias;
    t = max(t, t << rq.lk);
    const int rb = (int)__builtin_amdgcn_ubfe((unsigned int)t, (unsigned int)rq.sh_r, (unsigned int)rq.bw);
    return ((t << rq.sh_l) + rq.hm1 + rb) >> rq.sh_r;
}

struct Counters {
    unsigned long long absmax;   // max |t'| (stats mode)
    unsigned long long in_sat;   // layer 0 only: clamped input pixels
    unsigned long long sat;   // clamped outputs
    unsigned long long guard; // head-room violations
};

int y355_cu_count(void);       // engine.hip: compute units of the current device (cached per device), 256 on an MI355X in SPX mode
int y355_zero_counters(Counters *c, int n, hipStream_t s);    // engine.hip: a kernel launch, not hipMemsetAsync; returns the launch status (hipError_t)

// Two sets of per-forward counters used alternately: the fused front end of forward i (the first launch of a step) zeroes the
// set forward i + 1 will count into, so the steady state has NO counter-reset launch (round 5: a one-wave kernel per step,
// 4.5 us of a 298 us one-stream step).  Every launch of a stream runs behind the previous one, so the set being zeroed is idle:
// forward i - 1 (its last user) is complete, forward i counts into the other one, and a host read of "the last forward's
// counters" happens before the next forward is enqueued.  A forward without the fused front end zeroes its own set by a launch.
struct CounterSets {
    Counters *base = nullptr;     // [2][n]
    int n = 0, cur = 0;
    bool clean[2] = {false, false};
    Counters *set(int i) const { return base + (size_t)i * n; }
    // start a forward: the set it counts into; *need_zero: not zeroed by the previous forward's front end
    Counters *begin(bool *need_zero) {
        cur ^= 1;
        *need_zero = !clean[cur];
        clean[cur] = false;
        return set(cur);
    }
    // this forward's front end zeroes the other set (call when that launch is enqueued)
    Counters *other_zeroed_by_front() {
        clean[cur ^ 1] = true;
        return set(cur ^ 1);
    }
};

struct ConvParams {
    const int8_t *in;     // int8 NHWC with halo  [B][H+2][W+2][CIN]
    int8_t *out;          // int8 NHWC [B][Ho+2h][Wo+2h][cstride]
    const int8_t *w;      // fragment-packed weights
    const int *bias_t;    // [cout_pad]
    const long long *bias_w;  // [cout_pad] 64-bit copy for the wide epilogue
    Counters *ctr;
    int8_t *sink;         // >= 4 KiB scratch for masked-out stores (keeps store counts static)
    long long *raw;       // statistics mode: optional dump of t' as [B][H][W][cstride] (operator API)
    unsigned long long *stamps;   // diagnostic builds only: per-workgroup s_memtime stamps (or null)
    int B, H, W;          // input feature-map size (unpadded)
    int cstride;          // channels of the output buffer
    int out_halo;         // 1: output buffer carries a zero halo
    int tiles_x, tiles_y, nblk;
    Requant rq;
    int mode;             // 0 run, 1 statistics only
    int guard;            // evaluate the head-room guard
    // host side only (ring launcher): when set, the launch records the kernel's own start / end timestamps into these
    // events (hipExtLaunchKernelGGL) -- the duration rocprofv3 reports, without the gap to the neighbouring launches
    void *ev_start, *ev_stop;
    int grid_limit;       // host side only: persistent workgroups of a ring launch (0 = one per CU), Y355_OPT_RING_WORKGROUPS
    int xcd_share_log2;   // ring kernels: work items that read one input are walked by workgroups of one XCD (set by the launcher)
};

struct Conv1Params {
    const float *x;       // fp32 NCHW [B][3][H][W]
    const uint8_t *x_u8;  // or (x == nullptr) camera frames uint8 HWC BGR [B][H][W][3], normalised on the fly
    float nmean[3], nstd[3];  // BaseTransform constants per RGB channel (data/__init__.py:50)
    int8_t *out;          // int8 NHWC16 with halo [B][H/2+2][W/2+2][16]
    int out_pb;           // bytes per output pixel (0 = 16; wider: the extra bytes stay untouched)
    const int8_t *w;      // 64 lanes x 16 B fragment
    const int *bias_t;    // [16]
    const long long *bias_w;
    Counters *ctr;
    int B, H, W;
    int tiles_x, tiles_y;
    float in_scale;       // 2^sa[0]
    Requant rq;
    int mode;
    int guard;
};

// fused front end (front.hip): network input -> conv1 + pool1 -> conv2 + pool2
struct FrontParams {
    const float *x;       // fp32 NCHW [B][3][H][W]
    const uint8_t *x_u8;  // or (x == nullptr) uint8 HWC BGR frames [B][H][W][3]
    float nmean[3], nstd[3];
    int8_t *out;          // conv2's pooled output: int8 NHWC32 with halo [B][H/4+2][W/4+2][32]
    const int8_t *wf;     // 16 KiB of weight fragments (y355_pack_front)
    const int *bias1;     // [16]
    const int *bias2;     // [32]
    Counters *ctr;        // [0] conv1, [1] conv2
    unsigned long long *zero_next;   // or null: zero_n 64-bit words the first workgroup clears (the NEXT forward's counters, CounterSets)
    int zero_n;
    int B, H, W;
    int tiles_x, tiles_y;
    float in_scale;       // 2^sa[0]
    float in_thr;         // set by the launcher: 127.5 / in_scale
    int step_x, step_y, step_b;   // set by the launcher: the grid size split as (tiles_x, tiles_y, batch) digits (front.hip's tile walk)
    Requant rq1, rq2;
    unsigned long long *stamps;   // diagnostic builds only (-DFRONT_DIAG=1)
    void *ev_start, *ev_stop;     // host side only: see ConvParams
};
// fused front end of the bf16 nets (frontb.hip): fp32 NCHW -> conv(3 -> 16) + pool -> conv(16 -> 32) + pool -> bf16 NHWC
struct FrontBParams {
    const float *x;       // fp32 NCHW [B][3][H][W]
    char *out;            // conv2's pooled output: bf16 NHWC with halo [B][H/4+2][W/4+2][out_pb bytes], channels 0..31
    int out_pb;           // bytes per output pixel (64)
    const char *wf;       // 32 KiB of weight fragments (y355_pack_frontb)
    const float *bias1;   // [16]
    const float *bias2;   // [32]
    int B, H, W;
    int tiles_x, tiles_y;
    float slope1, slope2; // LeakyReLU slopes of the two layers, each in [0, 1] (the kernel takes max(m, m * slope))
    int step_x, step_y, step_b;   // set by the launcher: the grid size split as (tiles_x, tiles_y, batch) digits (the tile walk)
};
void y355_frontb_tiles(int H, int W, int *tx, int *ty);
void y355_pack_frontb(const float *w1 /*[16][3][3][3] or null*/, const float *w2 /*[32][16][3][3] or null*/, char *dst /*32768*/);
void y355_launch_frontb(const FrontBParams &p, hipStream_t s);
void y355_front_tiles(int H, int W, int *tx, int *ty);
void y355_pack_front(const int8_t *q_w1, const int8_t *q_w2, int8_t *dst /*16384; a null tensor leaves its part zero*/);
bool y355_front_eligible(const Requant &rq1, const Requant &rq2);
void y355_launch_front(const FrontParams &p, hipStream_t s);

template <typename T>
__device__ __forceinline__ T y355_rne_shift(T t, int sh) {
    if (sh > 0) {
        return (t + (((T)1 << (sh - 1)) - 1) + ((t >> sh) & 1)) >> sh;
    }
    return t * ((T)1 << (-sh));
}

template <typename T>
__device__ __forceinline__ T y355_pre(int acc, T bias, const Requant &rq) {
    T t = (T)acc * ((T)1 << rq.shl) + bias;
    if (rq.leaky) t = t >= 0 ? t * ((T)1 << rq.lk) : t * (T)rq.neg_mul;
    return t;
}

template <typename T> struct UnsignedOf { using type = unsigned int; };
template <> struct UnsignedOf<long long> { using type = unsigned long long; };
template <typename T>
__device__ __forceinline__ typename UnsignedOf<T>::type y355_uabs(T v) {
    return (typename UnsignedOf<T>::type)(v < 0 ? -v : v);
}

template <typename T>
__device__ __forceinline__ int y355_clamp8(T q) { return (int)min((T)127, max((T)-127, q)); }

// Blocks are dealt round-robin over the 8 XCDs; give each XCD a contiguous chunk of tile ids
// so neighbouring tiles (shared halo rows, same weights) hit the same L2.  Bijective for any n.
__device__ __forceinline__ int y355_xcd_remap(int bid, int n) {
    const int q = n >> 3, r = n & 7, x = bid & 7, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

__device__ __forceinline__ unsigned int y355_wave_max_u32(unsigned int v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned int)__shfl_xor((int)v, o, 64));
    return v;
}
__device__ __forceinline__ unsigned long long y355_wave_max_u64(unsigned long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const unsigned int lo = (unsigned int)__shfl_xor((int)(unsigned int)v, o, 64);
        const unsigned int hi = (unsigned int)__shfl_xor((int)(unsigned int)(v >> 32), o, 64);
        const unsigned long long w = ((unsigned long long)hi << 32) | lo;
        v = w > v ? w : v;
    }
    return v;
}

// conv3_1 (32 -> 64 channels at a quarter of the input resolution) tile geometry, shared by the generic kernel
// (conv3x3.hip, always 4 waves: GWM x WN); the weight packing depends on WN
#ifndef Y355_C31_TH
#define Y355_C31_TH 13
#define Y355_C31_TW 52
#define Y355_C31_WM 4
#define Y355_C31_WN 2
#endif
#define Y355_C31_GWM (4 / Y355_C31_WN)

// ---- host-side launch table --------------------------------------------------------------
struct ConvKernelInfo {
    int cin, bn, th, tw, pool, wm, wn;
    int nt, ks;
    size_t lds_bytes;
    void (*launch)(const ConvParams &p, int nblocks, hipStream_t s);
    int (*prepare)(void);     // one-time hipFuncSetAttribute
};

// pack q_w[cout][cin][3][3] into the fragment order kernel `ki` streams (host buffers).
void y355_pack_weights(const ConvKernelInfo &ki, const int8_t *q_w, int cout, int cin,
                       int cout_pad, int8_t *dst);
size_t y355_packed_bytes(const ConvKernelInfo &ki, int cout_pad);
const ConvKernelInfo *y355_conv_kernel(int id);
enum {
    Y355_K_CONV2 = 0, Y355_K_CONV3_1, Y355_K_CONV3_2, Y355_K_CONV4_1, Y355_K_CONV4_2,
    Y355_K_CONV5, Y355_K_CONV67, Y355_K_PRED,
    Y355_K_GEN16, Y355_K_GEN32, Y355_K_GEN64, Y355_K_GEN128, Y355_K_GEN256,
    Y355_K_GEN16P, Y355_K_GEN32P, Y355_K_GEN64P, Y355_K_GEN128P, Y355_K_GEN256P,
    Y355_K_COUNT
};

// conv3_1 .. conv4_2 with the weights in registers and the pixels as the MFMA's B operand (convpx.hip); p.w = y355_pack_px layout
bool y355_launch_conv_px(int kid, const ConvParams &p, hipStream_t s);
int y355_prepare_conv_px(void);
size_t y355_px_packed_bytes(int kid);           // 0: the layer has no such kernel
bool y355_pack_px(int kid, const int8_t *q_w, int cout, int cin, int8_t *dst);
// conv3_1 -> conv3_2 + pool in one launch (pxpair.hip): conv3_1's map stays in LDS
struct PairParams {
    const int8_t *in;     // conv3_1's input: int8 NHWC32 with halo [B][H+2][W+2][32]
    int8_t *out;          // conv3_2's pooled output: int8 NHWC64 with halo [B][H/2+2][W/2+2][64]
    const int8_t *w1;     // conv3_1's weights, y355_pack_px(Y355_K_CONV3_1) layout
    const int8_t *w2;     // conv3_2's weights, y355_pack_px(Y355_K_CONV3_2) layout
    const int *bias1;     // [64]
    const int *bias2;     // [64]
    Counters *ctr1, *ctr2;
    Requant rq1, rq2;
    int B, H, W;          // conv3_1's map (= its input's size, unpadded)
    void *ev_start, *ev_stop;     // host side only: see ConvParams
    int grid_limit;       // host side only: persistent workgroups per launch (0 = one per CU)
    unsigned long long *stamps;   // diagnostic builds only (-DPAIR_DIAG=1)
};
bool y355_pair3_eligible(const Requant &rq1, const Requant &rq2, int H, int W);
bool y355_launch_pair3(const PairParams &p, hipStream_t s);      // false: not eligible, run the two layers' own launches
int y355_prepare_pair3(void);
// deep-prefetch ring kernels (conv3x3_ring.hip), layers with >= 64 input channels
bool y355_launch_conv_ring(int kid, const ConvParams &p, hipStream_t s);
int y355_prepare_conv_ring(void);

void y355_launch_conv1(const Conv1Params &p, hipStream_t s);
void y355_conv1_tiles(int H, int W, int *tx, int *ty);
void y355_pack_conv1(const int8_t *q_w /*[16][3][3][3]*/, int8_t *dst /*1024*/);

// One prediction map of the detection head.  Channel layout [A obj | A*C cls | A*4 txtytwth],
// anchor-major (models/slim_yolo_v2.py:330-341, models/tiny_yolo_v3.py:202-222).
struct HeadLevel {
    const int8_t *pred;   // [B][Hs][Ws][cstride] int8 (value = q * dq) ...
    const float *pred_f;  // ... or fp32 (bf16 nets; pred == nullptr)
    int cstride;
    int Hs, Ws;
    float stride;         // cx = (sigmoid(tx) + gx) * stride
    float dq;             // 2^-sa_pred
    float anchors[32];    // (w, h) of this level's A anchors
};
struct HeadParams {
    HeadLevel lev[3];     // anchor index n: level 0 first, n = cell * A + a inside a level
    int nlev;
    int A, C;             // anchors per level, classes
    float wh_mul;         // w = exp(tw) * aw * wh_mul   (16: anchors in grid units; 1: pixels)
    int Hb, Wb;           // bin grid of the candidate sort
    int group_by_area;    // 0: candidates grouped by anchor type (bins = level 0's grid); 1: by area octave (<= 16 x 16 bins)
    int pairs_wgs;        // workgroups per image of the NMS pair walk: 0 = default (2); 1 while several handles share the GPU
    int cls_groups;       // set by y355_launch_head_nms: 1 = the groups are the CLASSES (per-class NMS never pairs two classes), else 0
    float in_w, in_h;     // network input size in pixels
    float conf_thresh, nms_thresh;
    float *cand_box;      // [B][N][4]
    float *cand_score;    // [B][N]
    int *cand_cls;        // [B][N]
    int max_det;
    float *out_box;
    float *out_score;
    int *out_cls;
    int *out_count;
};
#define Y355_NMS_CAP 4096   // anchors per image the NMS workspace is sized for
#define Y355_HEAD_EDGE_CAP 28672   // suppressing pairs per image the NMS edge list holds (more: the sorted fallback walk)
#define Y355_HEAD_MAXA 16
#define Y355_HEAD_MAXG 32   // candidate groups of the NMS sort: anchor types, area octaves, or -- heads with 3 .. 32 classes -- the classes
// head_nms.hip workspace, per image: cbox f32[CAP][4], cscore f32[CAP], ccls i32[CAP], corig i32[CAP],
// count i32, edges u32[Y355_HEAD_EDGE_CAP] (suppressing pairs), nedges i32[2] (count, overflow flag),
// binstart i32[CAP+8], astat f32[Y355_HEAD_MAXG][4], tiny i32[CAP], ntiny i32, ctype i32[CAP] (candidate group),
// dbox f32[CAP][4] / dscore f32[CAP] / dcls i32[CAP] (decode of every anchor).
// Heads with more than Y355_NMS_CAP anchors per image also need rbox f32[rstride][4], rscore f32[rstride], rcls i32[rstride]
// (raw decode, rstride >= anchors per image), rcount i32, ovf i32 (more than CAP anchors passed conf_thresh: zero it before
// a forward, check it after).
struct y355_head_ws { void *cbox, *cscore, *ccls, *corig, *count, *edges, *nedges, *binstart, *astat, *tiny, *ntiny, *ctype, *dbox, *dscore, *dcls;
                      void *rbox = nullptr, *rscore = nullptr, *rcls = nullptr, *rcount = nullptr, *ovf = nullptr; int rstride = 0; };
int y355_prepare_head(void);
// decode, candidate sort, pruned pair walk (edge list), rounds + output.  `mid` (optional) is recorded
// between the candidate sort and the pair walk.
// `kev` (optional): start / end events of the four launches decode, candidate sort, pair walk, rounds + output
void y355_launch_head_nms(const HeadParams &p, int batch, const y355_head_ws &ws, hipStream_t s, hipEvent_t mid,
                          hipEvent_t (*kev)[2] = nullptr);
void y355_launch_absmax(const float *x, size_t n, unsigned int *out_bits, hipStream_t s);
// uint8 HWC BGR frames -> fp32 NCHW RGB, BaseTransform arithmetic (data/__init__.py:30-56, test.py:79)
void y355_launch_normalize_u8(const uint8_t *frames, float *x, int B, int H, int W, const float *mean_rgb, const float *std_rgb,
                              hipStream_t s);

// ---- generic chunked conv (convg.hip): bf16 nets and the int8 layers conv3x3.hip cannot hold --
struct RequantG {
    int shl;       // t = acc * 2^shl + bias
    int sh;        // q = clamp(RNE(t' * 2^-sh))
    int lk;        // t' = t >= 0 ? t * 2^lk : t * neg_mul     (LeakyReLU slope neg_mul / 2^lk)
    int neg_mul;
    int narrow;    // 1: the whole epilogue fits 32 bits (host-checked bound): the 8-wave kernels take their 32-bit instantiation
    int split;     // narrow only: Requant::split (the negative branch's product in two halves)
};

struct ConvGParams {
    const char *in;           // NHWC with halo, in_pb bytes per pixel
    char *out;                // NHWC, out_pb bytes per pixel
    const char *w;            // fragment-packed weights (y355_convg_pack)
    const float *bias_f;      // bf16: [cout_pad]
    const long long *bias_w;  // int8: [cout_pad] pre-shifted
    Counters *ctr;            // int8: saturation counter (or null)
    int B, H, W;
    int in_pb, nchunks;       // bytes per input pixel; chunks of CHB bytes consumed
    int out_pb, out_off;      // bytes per output pixel of the buffer; byte offset of this conv's channel 0
    int out_halo, tiles_x, tiles_y, nblk;
    int taps;                 // 9 (3x3, pad 1) or 1 (1x1)
    float slope;              // bf16: y = x >= 0 ? x : slope * x
    int out_f32;              // bf16: store fp32 (prediction layers)
    RequantG rq;
    // bf16: residual added after the activation (backbone/darknet.py:36 `module(x) + x`), same layout as
    // `out` (halo, pixel pitch res_pb, byte offset res_off of channel 0); null = none
    const char *res;
    int res_pb, res_off;
    int grid_limit;           // host side only: persistent workgroups of a convr.hip launch (0 = one per CU), Y355_NET_OPT_WORKGROUPS
    int xcd_share_log2;       // convr.hip: work items that read one input are walked by workgroups of one XCD (set by the launcher)
};

struct Conv1FParams {
    const float *x;       // fp32 NCHW [B][3][H][W]
    char *out;            // bf16 NHWC16 with halo [B][H/2+2][W/2+2][16]
    const char *w;        // two 1 KiB fragments
    const float *bias;    // [16]
    int B, H, W, tiles_x, tiles_y;
    float slope;
};

struct ConvGInfo {
    int bf, chb, bn, th, tw, pool, wm, wn, nt;
    int stride;               // 1, or 2 (3x3 / pad 1 / stride 2: backbone/darknet.py:124-141)
    size_t lds_bytes;
    void (*launch)(const ConvGParams &p, int nblocks, hipStream_t s);
    int (*prepare)(void);
};
#define Y355_G_COUNT 11
const ConvGInfo *y355_convg_kernel(int bf, int id);
int y355_prepare_convg(void);
int y355_convg_select(int in_pb, int cout, int pool, int H, int W, int stride = 1, int batch_hint = 0);
int y355_convg_ksteps(const ConvGInfo &ki, int in_pb, int taps);
size_t y355_convg_packed_bytes(const ConvGInfo &ki, int in_pb, int taps, int cout_pad);
void y355_convg_pack(const ConvGInfo &ki, const float *w_f, const int8_t *w_q, int cout, int cin, int ksize,
                     int in_pb, int cout_pad, char *dst);
// thin 3x3 layers of the bf16 nets with the weights in registers (convpxb.hip): id from select (-1: none), weights fp32 [cout][cin][3][3]
int y355_prepare_convpxb(void);
int y355_convpxb_select(int in_pb, int cout, int pool);
size_t y355_convpxb_packed_bytes(int id);
bool y355_convpxb_pack(int id, const float *w, int cout, int cin, char *dst);
bool y355_launch_convpxb(int id, const ConvGParams &p, hipStream_t s);      // false: not available for this launch
// 3x3 layers of the generic nets on the LDS-DMA ring discipline (convr.hip); weights in y355_convg_pack order for (bn, wn, nt) below
struct Y355ConvRInfo { int bf, cinb, bn, th, tw, pool, wn, nt; };
int y355_prepare_convr(int device);
const Y355ConvRInfo *y355_convr_info(int rid);
int y355_convr_select(int bf, int in_pb, int cout_pad, int pool, int H, int W);
bool y355_launch_convr(int rid, const ConvGParams &p, int device, hipStream_t s);
bool y355_launch_pw_i8(const ConvGParams &p, hipStream_t s);        // 1x1, int8, 32-bit epilogue; weights packed for (bn 64, wn 1, nt 4)
void y355_conv1f_tiles(int H, int W, int *tx, int *ty);
void y355_launch_conv1f(const Conv1FParams &p, hipStream_t s);
void y355_pack_conv1f(const float *w, char *dst);

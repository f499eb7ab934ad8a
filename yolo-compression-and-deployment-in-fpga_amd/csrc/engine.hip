// yolo355 -- C ABI (include/yolo355.h): engine object, weight packing, layer schedule.
//
// Layer schedule = models/slim_yolo_v2.py:212-328 / c_embedding/yolo_forward.c:1202-1262:
//   conv1(3->16)+pool, conv2(16->32)+pool, conv3_1(32->64), conv3_2(64->64)+pool,
//   conv4_1(64->128), conv4_2(128->128)+pool, conv5(128->256), conv6, conv7, pred (no act).
#include "../../include/yolo355.h"
#include "y355_common.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

static thread_local std::string g_err;
int y355_fail(int code, const std::string &msg) {
    g_err = msg;
    return code;
}
static int fail(int code, const std::string &msg) { return y355_fail(code, msg); }
#define HIPCHK(expr)                                                                      \
    do {                                                                                  \
        hipError_t e_ = (expr);                                                           \
        if (e_ != hipSuccess)                                                             \
            return fail(Y355_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

extern "C" const char *y355_last_error(void) { return g_err.c_str(); }
extern "C" int y355_version(void) { return 2; }       // 2: round 6 (y355_pipeline_*, y355_calibrate, y355_conv_op, *_dev operators)

// throughput mode: a layer's launch runs on Y355_OPT_RING_WORKGROUPS x pct / 100 workgroups (A/B builds: -DY355_GL_TAB="{..10 values..}")
#ifndef Y355_GL_TAB
#define Y355_GL_TAB {100, 100, 100, 100, 100, 100, 100, 100, 100, 100}
#endif
static const int kGridLimitPct[10] = Y355_GL_TAB;
#ifndef Y355_TPUT_PAIRS_WGS
#define Y355_TPUT_PAIRS_WGS 1       // workgroups per image of the NMS pair walk while several handles share the GPU (alone: 2)
#endif
namespace {
struct LayerDef { int cin, cout, pool, leaky, kid; };
const LayerDef kLayers[10] = {
    {3, 16, 1, 1, -1},
    {16, 32, 1, 1, Y355_K_CONV2},
    {32, 64, 0, 1, Y355_K_CONV3_1},
    {64, 64, 1, 1, Y355_K_CONV3_2},
    {64, 128, 0, 1, Y355_K_CONV4_1},
    {128, 128, 1, 1, Y355_K_CONV4_2},
    {128, 256, 0, 1, Y355_K_CONV5},
    {256, 256, 0, 1, Y355_K_CONV67},
    {256, 256, 0, 1, Y355_K_CONV67},
    {256, 0, 0, 0, Y355_K_PRED},
};
const int kRetuneDefault[10] = {11, 10, 10, 11, 11, 10, 11, 11, 11, 10};  // yolo_forward.c:35

struct Layer {
    int cin = 0, cout = 0, cout_pad = 0, pool = 0, leaky = 0, kid = -1;
    int Hin = 0, Win = 0, Hout = 0, Wout = 0, halo = 1;
    int e_w = 0, e_b = 0;
    bool loaded = false, bias_dirty = true;
    std::vector<int32_t> q_b;
    int8_t *w_dev = nullptr;
    int8_t *wpx_dev = nullptr;      // conv3_1 .. conv4_2: the same weights in convpx.hip's fragment order
    long long wabs = 0;             // max over output channels of sum |q_w| (0: not loaded): the tight bound of |acc| / 127
    int *bias_dev = nullptr;
    long long *bias_w_dev = nullptr;
    int8_t *out_dev = nullptr;
    size_t out_bytes = 0;
    Requant rq{};
    int frac_bits = 0;
};

int kernels_prepared = 0;
}  // namespace
int y355_prepare_kernels() {
    if (kernels_prepared) return 0;
    for (int i = 0; i < Y355_K_COUNT; ++i) {
        int e = y355_conv_kernel(i)->prepare();
        if (e) return fail(Y355_EHIP, std::string("hipFuncSetAttribute: ") + hipGetErrorString((hipError_t)e));
    }
    if (int e = y355_prepare_head())
        return fail(Y355_EHIP, std::string("hipFuncSetAttribute(head): ") + hipGetErrorString((hipError_t)e));
    if (int e = y355_prepare_conv_ring())
        return fail(Y355_EHIP, std::string("hipFuncSetAttribute(ring): ") + hipGetErrorString((hipError_t)e));
    if (int e = y355_prepare_pair3())
        return fail(Y355_EHIP, std::string("hipFuncSetAttribute(pair3): ") + hipGetErrorString((hipError_t)e));
    if (int e = y355_prepare_conv_px())
        return fail(Y355_EHIP, std::string("hipFuncSetAttribute(px): ") + hipGetErrorString((hipError_t)e));
    if (int e = y355_prepare_convg())
        return fail(Y355_EHIP, std::string("hipFuncSetAttribute(convg): ") + hipGetErrorString((hipError_t)e));
    kernels_prepared = 1;
    return 0;
}
// compute units of the current device: the grid of the one-workgroup-per-CU launches (not the literal 256, so that a
// partitioned or smaller part does not serialise 157 KiB-of-LDS workgroups in waves)
int y355_cu_count(void) {
    static int cached[64] = {};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
    if (!cached[dev]) {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
        cached[dev] = n;
    }
    return cached[dev];
}
namespace {
int prepare_kernels() { return y355_prepare_kernels(); }

// Fill the integer epilogue of one layer.  Returns Y355_ERANGE when the int32 path could
// overflow for worst-case operands.
// act: 0 none (prediction layer), 1 LeakyReLU(0.125) (t' = t >= 0 ? 8 t : t, F' = F + 3), 2 ReLU (utils/modules.py:26 with
// leakyReLU=False: t' = max(t, 0), F' = F; runs on the generic kernels, whose epilogue honours neg_mul = 0)
// wabs: max over output channels of sum |q_w| when known (else 0: 127 per weight is assumed) -- only Requant::tmax_log2, the
// gate of the fp32-exact epilogues, uses this tight bound; `wide` keeps the operand-independent one
int make_requant(int cin_real, int sa_in, int e_w, int e_b, int sa_out, bool have_out, int act, int retune,
                 const int32_t *q_b, int cout, int cout_pad, Requant *rq, int *frac_bits, std::vector<int32_t> *bias_t,
                 std::vector<long long> *bias_w, long long wabs = 0) {
    const int F = std::max(sa_in + e_w, e_b);
    const int shl = F - sa_in - e_w, bshl = F - e_b;
    const int leaky = act == 1 ? 1 : 0;
    const int Fp = F + (leaky ? 3 : 0);
    if (shl > 30 || bshl > 40) return fail(Y355_ERANGE, "exponent gap too large for the fixed-point epilogue");
    long long bmax = 0;
    bias_t->assign(cout_pad, 0);
    bias_w->assign(cout_pad, 0);
    for (int c = 0; c < cout; ++c) {
        const long long v = (long long)q_b[c] * (1ll << bshl);
        bmax = std::max(bmax, std::llabs(v));
        (*bias_w)[c] = v;
        (*bias_t)[c] = (int32_t)v;     // only used when the 32-bit path is selected below
    }
    long long tmax = ((long long)127 * 127 * 9 * cin_real) * (1ll << shl) + bmax;
    if (leaky) tmax *= 8;
    int sh = have_out ? Fp - sa_out : 0;
    if (sh > 62) sh = 62;
    if (sh < -30) return fail(Y355_ERANGE, "output exponent too large");
    // worst-case magnitude through the rounding add / left shift
    long double lim = (long double)tmax;
    if (sh > 0) lim += std::ldexp(1.0L, sh - 1);
    if (sh < 0) lim = std::ldexp((long double)tmax, -sh);
    if (lim >= std::ldexp(1.0L, 62)) return fail(Y355_ERANGE, "fixed-point epilogue exceeds 62 bits");
    rq->wide = lim >= std::ldexp(1.0L, 30) ? 1 : 0;
    long long negsafe_t0 = 0;
    {
        const long long rowsum = wabs > 0 ? wabs : (long long)127 * 9 * cin_real;
        const long long t0 = (127 * rowsum) * (1ll << shl) + bmax;
        int n = 0;
        while (n < 62 && t0 >= (1ll << n)) ++n;
        rq->tmax_log2 = n;
        rq->negsafe = 0;
        negsafe_t0 = t0;
    }
    if (!rq->wide && sh > 31) sh = 31;
    rq->shl = shl;
    rq->sh = sh;
    rq->leaky = act ? 1 : 0;
    rq->lk = leaky ? 3 : 0;
    rq->neg_mul = act == 2 ? 0 : 1;
    rq->negsafe = std::ldexp((long double)negsafe_t0 * rq->neg_mul, -sh) <= 127.0L ? 1 : 0;
    rq->sh_l = sh < 0 ? -sh : 0;
    rq->sh_r = sh > 0 ? sh : 0;
    rq->hm1 = sh > 0 ? (int)((1ll << (sh - 1)) - 1) : 0;
    rq->bw = sh > 0 ? 1 : 0;
    rq->split = 0;
    int g = 15 + Fp - retune;
    rq->guard_log2 = g < 0 ? 0 : (g > 63 ? 63 : g);
    *frac_bits = Fp;
    return 0;
}
}  // namespace

// the integer epilogue of one stand-alone layer without requantisation (the operator objects of ops.hip)
int y355_op_requant(int cin, int sa_in, int e_w, int e_b, int act, const int32_t *q_b, int cout, int cout_pad, Requant *rq,
                    int *frac_bits, std::vector<int32_t> *bias_t, std::vector<long long> *bias_w) {
    return make_requant(cin, sa_in, e_w, e_b, 0, false, act, 10, q_b, cout, cout_pad, rq, frac_bits, bias_t, bias_w);
}

struct y355_engine {
    y355_config cfg{};
    hipStream_t stream = nullptr;
    bool own_stream = false;
    Layer L[10];
    int sa[11];
    bool sa_set[11];
    int retune[10];
    float trk_scale[11] = {};       // AveragedRangeTracker.scale / first_a of the 11 trackers (models/slim_yolo_v2.py:13-14), y355_calibrate
    int trk_first[11] = {};
    float nmean[3] = {0.485f, 0.456f, 0.406f};   // BaseTransform constants, RGB order (data/__init__.py:50 lists BGR)
    float nstd[3] = {0.229f, 0.224f, 0.225f};
    const uint8_t *x_u8 = nullptr;  // uint8 frames of the forward being enqueued (y355_forward_u8)
    // cv2.resize stage of BaseTransform (y355_forward_u8_resized): frames at the network size + coefficient tables
    uint8_t *rs_frames = nullptr;
    int *rs_tab = nullptr;          // [xofs W | xa 2W | yofs H | yb 2H]
    int rs_src_h = 0, rs_src_w = 0;
    int8_t *w0_dev = nullptr;       // conv1 fragment
    int8_t *wf_dev = nullptr;       // weight fragments of the fused front end (y355_pack_front)
    Counters *ctr_dev = nullptr;    // [10]: the set the last forward / layer run counted into (one of ctrs' two)
    CounterSets ctrs;
    unsigned int *absmax_dev = nullptr;
    int8_t *sink_dev = nullptr;
    unsigned long long *stamps_dev = nullptr;
    int Hs = 0, Ws = 0, N = 0;
    // head workspace
    y355_head_ws ws{};
    float *cand_box = nullptr, *cand_score = nullptr;
    int *cand_cls = nullptr;
    // host-call staging
    float *x_stage = nullptr;
    float *o_box = nullptr, *o_score = nullptr;
    int *o_cls = nullptr, *o_count = nullptr;
    int max_det = 0;
    int stamp_layer = -1;
    int profile = 0;
    int fuse_front = 1;             // conv1 + pool1 + conv2 + pool2 as one launch (front.hip) where eligible
    int fuse_pairs = 1;             // conv3_1 -> conv3_2 + pool3 as one launch (pxpair.hip) where eligible
    int l2_batch = 0;               // images of conv3_1's map that the last launches left valid in L[2].out_dev (the fused pair writes none)
    int l0_batch = 0;               // images of conv1's pooled map that the last launches left valid in L[0].out_dev (the fused
                                    // front end keeps that map on chip: 0 after a fused forward)
    int ring_wgs = 0;               // persistent workgroups per ring launch (0 = one per CU)
    hipEvent_t ev[Y355_NUM_TIMERS + 1];
    // kernel start / stop timestamps of the launches themselves (profile mode 2): 0..9 layers (the fused front end in slot 0),
    // 10 decode, 11 candidate sort, 12 pair walk, 13 rounds + output
    hipEvent_t kev[Y355_NUM_KERNEL_TIMERS][2];
    bool kev_set[Y355_NUM_KERNEL_TIMERS] = {};
    bool ev_ok = false;
    std::vector<void *> allocs;
};

static int dmalloc(y355_engine *h, void **p, size_t bytes, bool zero) {
    HIPCHK(hipMalloc(p, bytes ? bytes : 16));
    h->allocs.push_back(*p);
    if (zero) HIPCHK(hipMemset(*p, 0, bytes ? bytes : 16));
    return 0;
}

extern "C" void y355_destroy(y355_engine *h) {
    if (!h) return;
    (void)hipSetDevice(h->cfg.device_id);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (void *p : h->allocs) (void)hipFree(p);
    if (h->ev_ok) {
        for (auto &e : h->ev) (void)hipEventDestroy(e);
        for (auto &e : h->kev) { (void)hipEventDestroy(e[0]); (void)hipEventDestroy(e[1]); }
    }
    if (h->own_stream && h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

extern "C" int y355_create(const y355_config *cfg, y355_engine **out) {
    if (!cfg || !out) return fail(Y355_EINVAL, "null argument");
    if (cfg->height <= 0 || cfg->width <= 0 || cfg->height % 16 || cfg->width % 16)
        return fail(Y355_EINVAL, "input size must be a positive multiple of 16");
    if (cfg->num_anchors < 1 || cfg->num_anchors > Y355_MAX_ANCHORS || cfg->num_classes < 1)
        return fail(Y355_EINVAL, "bad anchors / classes");
    if (cfg->max_batch < 1) return fail(Y355_EINVAL, "max_batch < 1");
    const int predc = cfg->num_anchors * (5 + cfg->num_classes);
    if (predc > 256) return fail(Y355_EINVAL, "A*(5+C) > 256 not supported");
    const int Hs = cfg->height / 16, Ws = cfg->width / 16, N = Hs * Ws * cfg->num_anchors;
    if (N > Y355_NMS_CAP) return fail(Y355_EINVAL, "more than 4096 anchors per image not supported");
    HIPCHK(hipSetDevice(cfg->device_id));
    if (int e = prepare_kernels()) return e;
    y355_engine *h = new y355_engine();
    h->cfg = *cfg;
    h->Hs = Hs;
    h->Ws = Ws;
    h->N = N;
    h->max_det = (cfg->max_det <= 0 || cfg->max_det > N) ? N : cfg->max_det;
    for (int i = 0; i < 11; ++i) { h->sa[i] = 0; h->sa_set[i] = false; }
    for (int i = 0; i < 10; ++i) h->retune[i] = kRetuneDefault[i];
    if (!cfg->own_stream) {
        h->stream = (hipStream_t)cfg->stream;      // NULL = default stream (torch's default on ROCm)
    } else {
        if (hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking) != hipSuccess) {
            delete h;
            return fail(Y355_EHIP, "hipStreamCreate failed");
        }
        h->own_stream = true;
    }
    const int B = cfg->max_batch;
    int Hc = cfg->height, Wc = cfg->width;
    int rc = 0;
    for (int k = 0; k < 10 && !rc; ++k) {
        Layer &L = h->L[k];
        const LayerDef &d = kLayers[k];
        L.cin = d.cin;
        L.cout = d.cout ? d.cout : predc;
        L.pool = d.pool;
        L.leaky = d.leaky;
        L.kid = d.kid;
        L.Hin = Hc;
        L.Win = Wc;
        L.Hout = d.pool ? Hc / 2 : Hc;
        L.Wout = d.pool ? Wc / 2 : Wc;
        L.halo = (k == 9) ? 0 : 1;
        if (k == 0) L.cout_pad = 16;
        else {
            const int bn = y355_conv_kernel(L.kid)->bn;
            L.cout_pad = (L.cout + bn - 1) / bn * bn;
        }
        // slack rows so that clamped / masked tile reads of the next layer stay inside
        L.out_bytes = ((size_t)B * (L.Hout + 2 * L.halo) * (L.Wout + 2 * L.halo) + 64) * L.cout_pad;
        rc = dmalloc(h, (void **)&L.out_dev, L.out_bytes, true);
        if (!rc) rc = dmalloc(h, (void **)&L.bias_dev, sizeof(int) * L.cout_pad, true);
        if (!rc) rc = dmalloc(h, (void **)&L.bias_w_dev, sizeof(long long) * L.cout_pad, true);
        if (!rc && k > 0) rc = dmalloc(h, (void **)&L.w_dev, y355_packed_bytes(*y355_conv_kernel(L.kid), L.cout_pad), true);
        Hc = L.Hout;
        Wc = L.Wout;
    }
    const size_t cap = Y355_NMS_CAP;
    if (!rc) rc = dmalloc(h, (void **)&h->w0_dev, 1024, true);
    if (!rc) rc = dmalloc(h, (void **)&h->wf_dev, 16384, true);
    if (!rc) rc = dmalloc(h, (void **)&h->ctr_dev, sizeof(Counters) * 20, true);
    h->ctrs.base = h->ctr_dev;
    h->ctrs.n = 10;
    h->ctrs.clean[0] = h->ctrs.clean[1] = true;        // zeroed by the allocation
    if (!rc) rc = dmalloc(h, (void **)&h->absmax_dev, 16, true);
    if (!rc) rc = dmalloc(h, (void **)&h->sink_dev, 16384, true);
    if (!rc) rc = dmalloc(h, &h->ws.cbox, sizeof(float) * 4 * cap * B, false);
    if (!rc) rc = dmalloc(h, &h->ws.cscore, sizeof(float) * cap * B, false);
    if (!rc) rc = dmalloc(h, &h->ws.ccls, sizeof(int) * cap * B, false);
    if (!rc) rc = dmalloc(h, &h->ws.corig, sizeof(int) * cap * B, false);
    if (!rc) rc = dmalloc(h, &h->ws.count, sizeof(int) * B, true);
    if (!rc) rc = dmalloc(h, &h->ws.edges, sizeof(unsigned int) * (size_t)Y355_HEAD_EDGE_CAP * B, false);      // EDGE_CAP pairs per image
    if (!rc) rc = dmalloc(h, &h->ws.nedges, sizeof(int) * 2 * (size_t)B, true);
    if (!rc) rc = dmalloc(h, &h->ws.binstart, sizeof(int) * (cap + 8) * B, true);
    if (!rc) rc = dmalloc(h, &h->ws.astat, sizeof(float) * 4 * Y355_HEAD_MAXG * B, true);
    if (!rc) rc = dmalloc(h, &h->ws.tiny, sizeof(int) * cap * B, true);
    if (!rc) rc = dmalloc(h, &h->ws.ntiny, sizeof(int) * B, true);
    if (!rc) rc = dmalloc(h, &h->ws.dbox, sizeof(float) * 4 * cap * B, true);
    if (!rc) rc = dmalloc(h, &h->ws.dscore, sizeof(float) * cap * B, true);
    if (!rc) rc = dmalloc(h, &h->ws.dcls, sizeof(int) * cap * B, true);
    if (!rc) rc = dmalloc(h, &h->ws.ctype, sizeof(int) * cap * B, true);          // candidate groups
    if (!rc) rc = dmalloc(h, (void **)&h->cand_box, sizeof(float) * 4 * N * B, false);
    if (!rc) rc = dmalloc(h, (void **)&h->cand_score, sizeof(float) * N * B, false);
    if (!rc) rc = dmalloc(h, (void **)&h->cand_cls, sizeof(int) * N * B, false);
    if (!rc) rc = dmalloc(h, (void **)&h->o_box, sizeof(float) * 4 * h->max_det * B, false);
    if (!rc) rc = dmalloc(h, (void **)&h->o_score, sizeof(float) * h->max_det * B, false);
    if (!rc) rc = dmalloc(h, (void **)&h->o_cls, sizeof(int) * h->max_det * B, false);
    if (!rc) rc = dmalloc(h, (void **)&h->o_count, sizeof(int) * B, true);
    if (!rc) {
        bool ok = true;
        for (auto &e : h->ev) ok = ok && (hipEventCreate(&e) == hipSuccess);
        for (auto &e : h->kev) ok = ok && (hipEventCreate(&e[0]) == hipSuccess) && (hipEventCreate(&e[1]) == hipSuccess);
        h->ev_ok = ok;
        if (!ok) rc = fail(Y355_EHIP, "hipEventCreate failed");
    }
    if (rc) {
        std::string keep = g_err;
        y355_destroy(h);
        g_err = keep;
        return rc;
    }
    *out = h;
    return 0;
}

extern "C" int y355_set_option(y355_engine *h, int option, int value) {
    if (!h) return fail(Y355_EINVAL, "null engine");
    switch (option) {
    case Y355_OPT_FUSE_FRONT:
        if (value != 0 && value != 1) return fail(Y355_EINVAL, "Y355_OPT_FUSE_FRONT takes 0 or 1");
        h->fuse_front = value;
        return 0;
    case Y355_OPT_FUSE_PAIRS:
        if (value != 0 && value != 1) return fail(Y355_EINVAL, "Y355_OPT_FUSE_PAIRS takes 0 or 1");
        h->fuse_pairs = value;
        return 0;
    case Y355_OPT_RING_WORKGROUPS:
        if (value < 0 || value > 4096) return fail(Y355_EINVAL, "workgroups per launch out of range");
        h->ring_wgs = value;
        return 0;
    default: return fail(Y355_EINVAL, "unknown option");
    }
}

extern "C" int y355_set_thresholds(y355_engine *h, float conf, float nms) {
    if (!h) return fail(Y355_EINVAL, "null engine");
    h->cfg.conf_thresh = conf;
    h->cfg.nms_thresh = nms;
    return 0;
}

extern "C" int y355_load_layer(y355_engine *h, int idx, const int8_t *q_w, const int32_t *q_b, int cout, int cin,
                               int e_w, int e_b) {
    if (!h || !q_w || !q_b) return fail(Y355_EINVAL, "null argument");
    if (idx < 0 || idx >= 10) return fail(Y355_EINVAL, "layer index out of range");
    Layer &L = h->L[idx];
    if (cout != L.cout || cin != L.cin) {
        char buf[128];
        snprintf(buf, sizeof buf, "layer %d expects [%d,%d,3,3], got [%d,%d,3,3]", idx, L.cout, L.cin, cout, cin);
        return fail(Y355_EINVAL, buf);
    }
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    if (idx == 0) {
        int8_t frag[1024];
        y355_pack_conv1(q_w, frag);
        HIPCHK(hipMemcpy(h->w0_dev, frag, 1024, hipMemcpyHostToDevice));
        std::vector<int8_t> ff(16384);
        y355_pack_front(q_w, nullptr, ff.data());
        HIPCHK(hipMemcpy(h->wf_dev, ff.data(), 4096, hipMemcpyHostToDevice));
    } else {
        const ConvKernelInfo &ki = *y355_conv_kernel(L.kid);
        std::vector<int8_t> packed(y355_packed_bytes(ki, L.cout_pad));
        y355_pack_weights(ki, q_w, cout, cin, L.cout_pad, packed.data());
        HIPCHK(hipMemcpy(L.w_dev, packed.data(), packed.size(), hipMemcpyHostToDevice));
        if (idx == 1) {
            std::vector<int8_t> ff(16384);
            y355_pack_front(nullptr, q_w, ff.data());
            HIPCHK(hipMemcpy(h->wf_dev + 4096, ff.data() + 4096, 12288, hipMemcpyHostToDevice));
        }
    }
    {
        long long best = 0;
        const size_t per = (size_t)cin * 9;
        for (int c = 0; c < cout; ++c) {
            long long sum = 0;
            for (size_t i = 0; i < per; ++i) sum += std::abs((int)q_w[(size_t)c * per + i]);
            best = std::max(best, sum);
        }
        L.wabs = best;
    }
    if (const size_t pxb = idx > 0 ? y355_px_packed_bytes(L.kid) : 0) {
        std::vector<int8_t> px(pxb);
        if (y355_pack_px(L.kid, q_w, cout, cin, px.data())) {
            if (!L.wpx_dev) {
                if (int rc = dmalloc(h, (void **)&L.wpx_dev, px.size(), false)) return rc;
            }
            HIPCHK(hipMemcpy(L.wpx_dev, px.data(), px.size(), hipMemcpyHostToDevice));
        }
    }
    L.q_b.assign(q_b, q_b + cout);
    L.e_w = e_w;
    L.e_b = e_b;
    L.loaded = true;
    L.bias_dirty = true;
    return 0;
}

extern "C" int y355_set_act_exponents(y355_engine *h, const int32_t *sa) {
    if (!h || !sa) return fail(Y355_EINVAL, "null argument");
    for (int i = 0; i < 11; ++i)
        if (sa[i] < -64 || sa[i] > 64) return fail(Y355_EINVAL, "activation exponent out of range");
    // only what changes makes a layer's epilogue stale (callers set the same frozen exponents before every forward: that must
    // not cost ten bias uploads and stream synchronisations per call)
    for (int i = 0; i < 11; ++i) {
        if (h->sa_set[i] && h->sa[i] == sa[i]) continue;
        h->sa[i] = sa[i];
        h->sa_set[i] = true;
        if (i < 10) h->L[i].bias_dirty = true;
        if (i > 0) h->L[i - 1].bias_dirty = true;
    }
    return 0;
}

static int set_one_exponent(y355_engine *h, int i, int v) {
    if (h->sa_set[i] && h->sa[i] == v) return 0;
    h->sa[i] = v;
    h->sa_set[i] = true;
    if (i < 10) h->L[i].bias_dirty = true;
    if (i > 0) h->L[i - 1].bias_dirty = true;
    return 0;
}

extern "C" int y355_set_act_exponent(y355_engine *h, int i, int v) {
    if (!h || i < 0 || i > 10 || v < -64 || v > 64) return fail(Y355_EINVAL, "bad exponent");
    return set_one_exponent(h, i, v);
}

extern "C" int y355_get_act_exponents(y355_engine *h, int32_t *sa) {
    if (!h || !sa) return fail(Y355_EINVAL, "null argument");
    for (int i = 0; i < 11; ++i) sa[i] = h->sa[i];
    return 0;
}

extern "C" int y355_set_retune(y355_engine *h, const int32_t *r) {
    if (!h || !r) return fail(Y355_EINVAL, "null argument");
    for (int i = 0; i < 10; ++i) {
        if (h->retune[i] == r[i]) continue;
        h->retune[i] = r[i];
        h->L[i].bias_dirty = true;
    }
    return 0;
}

// (re)derive the integer epilogue of layer k and upload its pre-shifted biases
static int refresh_layer(y355_engine *h, int k, bool need_out) {
    Layer &L = h->L[k];
    if (!L.loaded) return fail(Y355_ENOTREADY, "layer weights not loaded");
    if (!h->sa_set[k]) return fail(Y355_ENOTREADY, "input activation exponent not set (calibrate first)");
    if (need_out && !h->sa_set[k + 1]) return fail(Y355_ENOTREADY, "output activation exponent not set (calibrate first)");
    if (!L.bias_dirty) return 0;
    std::vector<int32_t> bt;
    std::vector<long long> bw;
    int rc = make_requant(L.cin, h->sa[k], L.e_w, L.e_b, h->sa[k + 1], h->sa_set[k + 1], L.leaky, h->retune[k],
                          L.q_b.data(), L.cout, L.cout_pad, &L.rq, &L.frac_bits, &bt, &bw, L.wabs);
    if (rc) return rc;
    HIPCHK(hipMemcpyAsync(L.bias_dev, bt.data(), sizeof(int) * L.cout_pad, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipMemcpyAsync(L.bias_w_dev, bw.data(), sizeof(long long) * L.cout_pad, hipMemcpyHostToDevice, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));   // bt is a stack-owned vector
    L.bias_dirty = !h->sa_set[k + 1];
    return 0;
}

static int launch_layer(y355_engine *h, int k, int B, int mode, int guard, const float *x_dev) {
    Layer &L = h->L[k];
    if (k == 0) {
        Conv1Params p{};
        p.x = x_dev;
        p.x_u8 = x_dev ? nullptr : h->x_u8;
        for (int c = 0; c < 3; ++c) { p.nmean[c] = h->nmean[c]; p.nstd[c] = h->nstd[c]; }
        p.out = L.out_dev;
        p.w = h->w0_dev;
        p.bias_t = L.bias_dev;
        p.bias_w = L.bias_w_dev;
        p.ctr = h->ctr_dev;
        p.B = B;
        p.H = L.Hin;
        p.W = L.Win;
        y355_conv1_tiles(L.Hin, L.Win, &p.tiles_x, &p.tiles_y);
        p.in_scale = std::ldexp(1.0f, h->sa[0]);
        p.rq = L.rq;
        p.mode = mode;
        p.guard = guard;
        y355_launch_conv1(p, h->stream);
        if (mode == 0) h->l0_batch = B;
    } else {
        const ConvKernelInfo &ki = *y355_conv_kernel(L.kid);
        ConvParams p{};
        p.in = h->L[k - 1].out_dev;
        p.out = L.out_dev;
        p.w = L.w_dev;
        p.bias_t = L.bias_dev;
        p.bias_w = L.bias_w_dev;
        p.ctr = h->ctr_dev + k;
        p.sink = h->sink_dev;
        p.stamps = (h->stamp_layer == k) ? h->stamps_dev : nullptr;
        h->kev_set[k] = false;
        if (h->profile == 2 && mode == 0) { p.ev_start = h->kev[k][0]; p.ev_stop = h->kev[k][1]; }
        p.grid_limit = h->ring_wgs * kGridLimitPct[k] / 100;
        p.B = B;
        p.H = L.Hin;
        p.W = L.Win;
        p.cstride = L.cout_pad;
        p.out_halo = L.halo;
        p.tiles_x = (L.Win + ki.tw - 1) / ki.tw;
        p.tiles_y = (L.Hin + ki.th - 1) / ki.th;
        p.nblk = L.cout_pad / ki.bn;
        p.rq = L.rq;
        p.mode = mode;
        p.guard = guard;
        // one production kernel per layer, then ONE fallback family (conv3x3.hip: statistics mode, head-room guard, 64-bit
        // epilogue, any shape).  conv3_1 .. conv4_2: weights in registers, pixels as the B operand (convpx.hip); conv3_2 .. pred:
        // LDS-DMA rings (conv3x3_ring.hip); conv2 (only when the fused front end is off) and whatever those two refuse: conv3x3.hip.
        // (Round 4: the resident-weight family conv3x3_v2.hip is gone -- it only ever ran where the fused front end or convpx refused.)
#ifdef Y355_EXPERIMENTS
        static const int no_ring_mask = getenv("Y355_NO_RING_MASK") ? atoi(getenv("Y355_NO_RING_MASK")) : 0;
        static const int no_px_mask = getenv("Y355_NO_PX_MASK") ? atoi(getenv("Y355_NO_PX_MASK")) : 0;
#else
        constexpr int no_ring_mask = 0, no_px_mask = 0;
#endif
        if (k == 2 && mode == 0) h->l2_batch = B;
        if (L.wpx_dev && !((no_px_mask >> k) & 1)) {
            ConvParams q = p;
            q.w = L.wpx_dev;
            if (y355_launch_conv_px(L.kid, q, h->stream)) {
                HIPCHK(hipGetLastError());
                h->kev_set[k] = p.ev_start != nullptr;
                return 0;
            }
        }
        if (!((no_ring_mask >> k) & 1) && y355_launch_conv_ring(L.kid, p, h->stream)) {
            HIPCHK(hipGetLastError());
            h->kev_set[k] = p.ev_start != nullptr;
            return 0;
        }
        ki.launch(p, p.tiles_x * p.tiles_y * p.nblk * B, h->stream);
    }
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int y355_input_absmax(y355_engine *h, const float *x_dev, int batch, float *out_max) {
    if (!h || !x_dev || !out_max) return fail(Y355_EINVAL, "null argument");
    if (batch < 1 || batch > h->cfg.max_batch) return fail(Y355_EINVAL, "batch out of range");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipMemsetAsync(h->absmax_dev, 0, 16, h->stream));
    y355_launch_absmax(x_dev, (size_t)batch * 3 * h->cfg.height * h->cfg.width, h->absmax_dev, h->stream);
    HIPCHK(hipGetLastError());
    unsigned int bits = 0;
    HIPCHK(hipMemcpyAsync(&bits, h->absmax_dev, 4, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    memcpy(out_max, &bits, 4);
    return 0;
}

extern "C" int y355_run_layer(y355_engine *h, int idx, int batch, int mode, const float *x_dev) {
    if (!h) return fail(Y355_EINVAL, "null engine");
    if (idx < 0 || idx >= 10 || (mode != 0 && mode != 1)) return fail(Y355_EINVAL, "bad layer / mode");
    if (batch < 1 || batch > h->cfg.max_batch) return fail(Y355_EINVAL, "batch out of range");
    if (idx == 0 && !x_dev) return fail(Y355_EINVAL, "layer 0 needs the network input");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    if (int rc = refresh_layer(h, idx, mode == 0)) return rc;
    HIPCHK(hipMemsetAsync(h->ctr_dev + idx, 0, sizeof(Counters), h->stream));
    return launch_layer(h, idx, batch, mode, 1, x_dev);
}

extern "C" int y355_layer_stats_get(y355_engine *h, int idx, y355_layer_stats *out) {
    if (!h || !out || idx < 0 || idx >= 10) return fail(Y355_EINVAL, "bad argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    Counters c;
    HIPCHK(hipMemcpyAsync(&c, h->ctr_dev + idx, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    out->absmax_t = (int64_t)c.absmax;
    out->frac_bits = h->L[idx].frac_bits;
    out->reserved = (int32_t)c.in_sat;
    out->saturated = (int64_t)c.sat + (idx == 0 ? c.in_sat : 0);
    out->guard = (int64_t)c.guard;
    return 0;
}

// ---- AveragedRangeTracker.quantize_activation's state machine (models/slim_yolo_v2.py:16-38) for the 11 trackers of the
// path, in the reference's float32 arithmetic:
//   _max = activation.abs().max(); scale = 127 / _max        -- Python int / tensor = Tensor.__rtruediv__ = reciprocal() * 127: TWO
//                                                               fp32 roundings, reproduced here
//   first call ever (first_a == 0, even when frozen, :25-27): first_a = 1, self.scale += scale
//   frozen (:28-29): unchanged;  else (:30-31) self.scale = self.scale * (1 - momentum) + scale * momentum, the Python doubles
//   1 - momentum and momentum each rounded to fp32 by the tensor-scalar ops
//   exponent = floor(log2(self.scale)) (:33), log2 in fp32
static int tracker_update(float *scale, int *first, int i, float max_abs, int freeze, double momentum, int *exp_out) {
    const float s = (1.0f / max_abs) * 127.0f;
    if (!*first) {
        *first = 1;
        *scale = *scale + s;
    } else if (!freeze) {
        const float keep = (float)(1.0 - momentum), mom = (float)momentum;
        *scale = *scale * keep + s * mom;
    }
    const float sc = *scale;
    if (!(sc > 0.f) || !std::isfinite(sc)) {
        char buf[160];
        snprintf(buf, sizeof buf, "tracker %d: scale %g is not a positive finite number (max|activation| = %g)", i, (double)sc, (double)max_abs);
        return fail(Y355_ERANGE, buf);
    }
    const int e = (int)std::floor(std::log2(sc));
    if (e < -64 || e > 64) return fail(Y355_ERANGE, "tracker exponent out of range");
    *exp_out = e;
    return 0;
}

// the update of ONE tracker as a host utility (no GPU): the arithmetic y355_calibrate applies, testable against torch on the CPU
extern "C" int y355_tracker_step(float *scale, int32_t *first_a, float max_abs, int freeze, double momentum, int32_t *exponent) {
    if (!scale || !first_a || !exponent) return fail(Y355_EINVAL, "null argument");
    int first = *first_a ? 1 : 0, e = 0;
    const int rc = tracker_update(scale, &first, 0, max_abs, freeze, momentum, &e);
    *first_a = first;
    if (!rc) *exponent = e;
    return rc;
}

extern "C" int y355_set_trackers(y355_engine *h, const float *scale, const int32_t *first_a) {
    if (!h || !scale || !first_a) return fail(Y355_EINVAL, "null argument");
    for (int i = 0; i < 11; ++i) {
        h->trk_scale[i] = scale[i];
        h->trk_first[i] = first_a[i] ? 1 : 0;
    }
    return 0;
}

extern "C" int y355_get_trackers(y355_engine *h, float *scale, int32_t *first_a) {
    if (!h || !scale || !first_a) return fail(Y355_EINVAL, "null argument");
    for (int i = 0; i < 11; ++i) {
        scale[i] = h->trk_scale[i];
        first_a[i] = h->trk_first[i];
    }
    return 0;
}

// One calibration step on a batch: what forward(x, quantization=True) does to the trackers (:212-328), layer by layer on the
// GPU -- every tracker sees max|activation| of its layer computed with the exponents just updated in front of it, and the
// layer then runs for real so that the next one reads the reference's fake-quantised map.  Leaves the engine's exponents set.
extern "C" int y355_calibrate(y355_engine *h, const float *x_dev, int batch, int freeze, double momentum, int32_t *sa_out,
                              float *max_out) {
    if (!h || !x_dev) return fail(Y355_EINVAL, "null argument");
    if (batch < 1 || batch > h->cfg.max_batch) return fail(Y355_EINVAL, "batch out of range");
    if (!(momentum >= 0.0 && momentum <= 1.0)) return fail(Y355_EINVAL, "momentum outside [0, 1]");
    float m = 0.f;
    if (int rc = y355_input_absmax(h, x_dev, batch, &m)) return rc;
    int e = 0;
    if (int rc = tracker_update(&h->trk_scale[0], &h->trk_first[0], 0, m, freeze, momentum, &e)) return rc;
    if (max_out) max_out[0] = m;
    if (sa_out) sa_out[0] = e;
    set_one_exponent(h, 0, e);
    for (int k = 0; k < 10; ++k) {
        const float *xp = k == 0 ? x_dev : nullptr;
        if (int rc = y355_run_layer(h, k, batch, 1, xp)) return rc;
        y355_layer_stats st;
        if (int rc = y355_layer_stats_get(h, k, &st)) return rc;
        // max|y| as the fp32 value the reference's activation.abs().max() returns: t' / 2^F' is exact in fp32 wherever the
        // reference's own fp32 convolution is
        const float ymax = (float)st.absmax_t * std::ldexp(1.0f, -st.frac_bits);
        if (int rc = tracker_update(&h->trk_scale[k + 1], &h->trk_first[k + 1], k + 1, ymax, freeze, momentum, &e)) return rc;
        if (max_out) max_out[k + 1] = ymax;
        if (sa_out) sa_out[k + 1] = e;
        set_one_exponent(h, k + 1, e);
        if (int rc = y355_run_layer(h, k, batch, 0, xp)) return rc;
    }
    return 0;
}

extern "C" int y355_get_feature(y355_engine *h, int idx, int batch, int8_t *dst) {
    if (!h || !dst || idx < 0 || idx >= 10) return fail(Y355_EINVAL, "bad argument");
    if (batch < 1 || batch > h->cfg.max_batch) return fail(Y355_EINVAL, "batch out of range");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    if (idx == 0 && batch > h->l0_batch)
        return fail(Y355_ENOTREADY, "conv1's map of the last forward was not written (fused front end): "
                                    "run with y355_set_option(h, Y355_OPT_FUSE_FRONT, 0) to tap it");
    if (idx == 2 && batch > h->l2_batch)
        return fail(Y355_ENOTREADY, "conv3_1's map of the last forward was not written (fused conv3_1 -> conv3_2 + pool): "
                                    "run with y355_set_option(h, Y355_OPT_FUSE_PAIRS, 0) to tap it");
    const Layer &L = h->L[idx];
    const int Hp = L.Hout + 2 * L.halo, Wp = L.Wout + 2 * L.halo, CS = L.cout_pad;
    std::vector<int8_t> tmp((size_t)batch * Hp * Wp * CS);
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(tmp.data(), L.out_dev, tmp.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < batch; ++b)
        for (int c = 0; c < L.cout; ++c)
            for (int y = 0; y < L.Hout; ++y)
                for (int x = 0; x < L.Wout; ++x)
                    dst[(((size_t)b * L.cout + c) * L.Hout + y) * L.Wout + x] =
                        tmp[(((size_t)b * Hp + y + L.halo) * Wp + x + L.halo) * CS + c];
    return 0;
}

static HeadParams head_params(y355_engine *h, int sa_pred, float *ob, float *os, int *oc, int *on) {
    HeadParams p{};
    const Layer &L = h->L[9];
    p.nlev = 1;
    HeadLevel &lv = p.lev[0];
    lv.pred = L.out_dev;
    lv.pred_f = nullptr;
    lv.cstride = L.cout_pad;
    lv.Hs = h->Hs;
    lv.Ws = h->Ws;
    lv.stride = 16.0f;                       // models/slim_yolo_v2.py:52
    lv.dq = std::ldexp(1.0f, -sa_pred);
    for (int i = 0; i < 2 * h->cfg.num_anchors; ++i) lv.anchors[i] = h->cfg.anchors[i];
    p.A = h->cfg.num_anchors;
    p.C = h->cfg.num_classes;
    p.wh_mul = 16.0f;                        // anchors in grid units (:126)
    // int8 logits keep exp(tw) in a narrow range: the anchor is a good size class (measured: pairs 52 us
    // vs 80 us with area octaves on the benchmark batch)
    p.group_by_area = 0;
    p.pairs_wgs = h->ring_wgs > 0 ? Y355_TPUT_PAIRS_WGS : 0;    // throughput mode (Y355_OPT_RING_WORKGROUPS set): the pair walk holds one CU per image
    p.Hb = h->Hs;
    p.Wb = h->Ws;
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

// conv1 + pool1 + conv2 + pool2 in one launch (front.hip); the 16-channel map never reaches HBM
static int launch_front(y355_engine *h, int B, const float *x_dev) {
    Layer &L0 = h->L[0], &L1 = h->L[1];
    FrontParams p{};
    p.x = x_dev;
    p.x_u8 = x_dev ? nullptr : h->x_u8;
    for (int c = 0; c < 3; ++c) { p.nmean[c] = h->nmean[c]; p.nstd[c] = h->nstd[c]; }
    p.out = L1.out_dev;
    p.wf = h->wf_dev;
    p.bias1 = L0.bias_dev;
    p.bias2 = L1.bias_dev;
    p.ctr = h->ctr_dev;
    p.zero_next = (unsigned long long *)h->ctrs.other_zeroed_by_front();
    p.zero_n = 10 * (int)(sizeof(Counters) / 8);
    p.B = B;
    p.H = L0.Hin;
    p.W = L0.Win;
    y355_front_tiles(L0.Hin, L0.Win, &p.tiles_x, &p.tiles_y);
    p.in_scale = std::ldexp(1.0f, h->sa[0]);
    p.rq1 = L0.rq;
    p.rq2 = L1.rq;
    p.stamps = (h->stamp_layer == 0) ? h->stamps_dev : nullptr;
    h->kev_set[0] = h->kev_set[1] = false;
    if (h->profile == 2) { p.ev_start = h->kev[0][0]; p.ev_stop = h->kev[0][1]; h->kev_set[0] = true; }
    y355_launch_front(p, h->stream);
    HIPCHK(hipGetLastError());
    return 0;
}

// The per-forward reset of the saturation counters as an ordinary kernel launch instead of hipMemsetAsync (whose fill goes down
// the runtime's blit path).  Measured equal to the memset in the three-handle regime (299.7 k against 300.0 k img/s); under
// rocprofv3 the memset shows a 280 us gap in front of it, which is the profiler's own per-dispatch cost, not the runtime's.
__global__ void y355_zero_u64_kernel(unsigned long long *p, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) p[i] = 0ull;
}
int y355_zero_counters(Counters *c, int n, hipStream_t s) {
    hipLaunchKernelGGL(y355_zero_u64_kernel, dim3(1), dim3(64), 0, s, (unsigned long long *)c, n * (int)(sizeof(Counters) / 8));
    return (int)hipGetLastError();                   // a failed launch must not leave the previous forward's counts in place
}

// conv3_1 -> conv3_2 + pool3 as one launch (pxpair.hip): 1 = launched, 0 = not eligible (the caller runs the two layers), < 0 error
static int launch_pair3(y355_engine *h, int B) {
    Layer &A = h->L[2], &Bl = h->L[3];
    if (!A.wpx_dev || !Bl.wpx_dev || A.cin != 32 || A.cout_pad != 64 || Bl.cout_pad != 64 || !Bl.pool || A.pool) return 0;
#ifdef Y355_EXPERIMENTS
    static const bool no_pair = getenv("Y355_NO_PAIR3") != nullptr;
    if (no_pair) return 0;
#endif
    PairParams p{};
    p.in = h->L[1].out_dev;
    p.out = Bl.out_dev;
    p.w1 = A.wpx_dev;
    p.w2 = Bl.wpx_dev;
    p.bias1 = A.bias_dev;
    p.bias2 = Bl.bias_dev;
    p.ctr1 = h->ctr_dev + 2;
    p.ctr2 = h->ctr_dev + 3;
    p.rq1 = A.rq;
    p.rq2 = Bl.rq;
    p.B = B;
    p.H = A.Hin;
    p.W = A.Win;
    p.grid_limit = h->ring_wgs * kGridLimitPct[2] / 100;
    p.stamps = (h->stamp_layer == 2) ? h->stamps_dev : nullptr;
    h->kev_set[2] = h->kev_set[3] = false;
    if (h->profile == 2) { p.ev_start = h->kev[2][0]; p.ev_stop = h->kev[2][1]; }
    if (!y355_launch_pair3(p, h->stream)) return 0;
    HIPCHK(hipGetLastError());
    h->kev_set[2] = p.ev_start != nullptr;
    h->l2_batch = 0;
    return 1;
}

// enqueue one forward on `s` (refresh_layer must have run)
static int enqueue_forward(y355_engine *h, const float *x_dev, int batch, int flags, float *boxes_dev, float *scores_dev,
                           int32_t *cls_dev, int32_t *count_dev, bool prof) {
    bool need_zero = false;
    h->ctr_dev = h->ctrs.begin(&need_zero);          // steady state: zeroed by the previous forward's front end, no launch here
    if (need_zero) HIPCHK((hipError_t)y355_zero_counters(h->ctr_dev, 10, h->stream));
    const int guard = (flags & Y355_F_GUARD) ? 1 : 0;
    // the fused front end covers the 32-bit epilogue without the head-room guard; conv2's packed weights must be the
    // resident-weight layout (one n-block of 32 channels), which they are for this network
    const bool fused = h->fuse_front && !guard && y355_front_eligible(h->L[0].rq, h->L[1].rq) && h->L[1].cout_pad == 32;
    bool pair3 = false;
    for (int k = 0; k < 10; ++k) {
        if (prof) HIPCHK(hipEventRecord(h->ev[k], h->stream));
        if (fused && k == 0) {
            h->l0_batch = 0;
            if (int rc = launch_front(h, batch, x_dev)) return rc;
            continue;
        }
        if (fused && k == 1) continue;                    // timer slot 1 reads ~0: slot 0 holds conv1 + conv2
        if (k == 2 && h->fuse_pairs && !guard) {          // conv3_1 -> conv3_2 + pool3 in one launch; slot 3 then reads ~0
            const int rc = launch_pair3(h, batch);
            if (rc < 0) return rc;
            pair3 = rc == 1;
            if (pair3) continue;
        }
        if (pair3 && k == 3) continue;
        if (int rc = launch_layer(h, k, batch, 0, guard, x_dev)) return rc;
    }
    if (prof) HIPCHK(hipEventRecord(h->ev[10], h->stream));
    HeadParams hp = head_params(h, h->sa[10], boxes_dev, scores_dev, cls_dev, count_dev);
    if (!(flags & Y355_F_TAP)) { hp.cand_box = nullptr; hp.cand_score = nullptr; hp.cand_cls = nullptr; }
    const bool kprof = prof && h->profile == 2;
    for (int i = 10; i < Y355_NUM_KERNEL_TIMERS; ++i) h->kev_set[i] = kprof;
    y355_launch_head_nms(hp, batch, h->ws, h->stream, prof ? h->ev[11] : nullptr, kprof ? &h->kev[10] : nullptr);
    HIPCHK(hipGetLastError());
    if (prof) HIPCHK(hipEventRecord(h->ev[12], h->stream));
    return 0;
}

extern "C" int y355_forward(y355_engine *h, const float *x_dev, int batch, int flags, float *boxes_dev,
                            float *scores_dev, int32_t *cls_dev, int32_t *count_dev) {
    if (!h || !x_dev || !boxes_dev || !scores_dev || !cls_dev || !count_dev) return fail(Y355_EINVAL, "null argument");
    if (batch < 1 || batch > h->cfg.max_batch) return fail(Y355_EINVAL, "batch out of range");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    for (int k = 0; k < 10; ++k)
        if (int rc = refresh_layer(h, k, true)) return rc;
    const bool prof = h->profile != 0;
    // (rounds 2-5 carried a hipGraph replay of the step's launches behind -DY355_EXPERIMENTS: measured, no gain; removed with the
    // alternating counter sets, which a captured graph would freeze)
    return enqueue_forward(h, x_dev, batch, flags, boxes_dev, scores_dev, cls_dev, count_dev, prof);
}

// BaseTransform constants of the uint8 path, in the reference's BGR order (data/__init__.py:50)
extern "C" int y355_set_normalization(y355_engine *h, const float *mean_bgr, const float *std_bgr) {
    if (!h || !mean_bgr || !std_bgr) return fail(Y355_EINVAL, "null argument");
    for (int c = 0; c < 3; ++c) {
        if (!(std_bgr[2 - c] > 0.f)) return fail(Y355_EINVAL, "std must be positive");
        h->nmean[c] = mean_bgr[2 - c];
        h->nstd[c] = std_bgr[2 - c];
    }
    return 0;
}

// The step in front of the path: frames as cv2 delivers them, uint8 HWC BGR [B][H][W][3]
// already at the network size; BaseTransform + BGR->RGB + HWC->CHW (data/__init__.py:30-56, test.py:79)
// are fused into the first layer's load (4x fewer input bytes than the fp32 tensor).  Same outputs as
// y355_forward on the normalised tensor, bit for bit.
extern "C" int y355_forward_u8(y355_engine *h, const uint8_t *frames_dev, int batch, int flags, float *boxes_dev,
                               float *scores_dev, int32_t *cls_dev, int32_t *count_dev) {
    if (!h || !frames_dev || !boxes_dev || !scores_dev || !cls_dev || !count_dev) return fail(Y355_EINVAL, "null argument");
    if (batch < 1 || batch > h->cfg.max_batch) return fail(Y355_EINVAL, "batch out of range");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    for (int k = 0; k < 10; ++k)
        if (int rc = refresh_layer(h, k, true)) return rc;
    const bool fused = !h->L[0].rq.wide && !(flags & Y355_F_GUARD);
    if (!fused) {
        // 64-bit epilogue / guard runs take the fp32 tensor: normalise into the staging buffer first
        const size_t xin = (size_t)3 * h->cfg.height * h->cfg.width;
        if (!h->x_stage) {
            if (int rc = dmalloc(h, (void **)&h->x_stage, sizeof(float) * xin * h->cfg.max_batch, false)) return rc;
        }
        y355_launch_normalize_u8(frames_dev, h->x_stage, batch, h->cfg.height, h->cfg.width, h->nmean, h->nstd, h->stream);
        HIPCHK(hipGetLastError());
        return enqueue_forward(h, h->x_stage, batch, flags, boxes_dev, scores_dev, cls_dev, count_dev, h->profile != 0);
    }
    h->x_u8 = frames_dev;
    const int rc = enqueue_forward(h, nullptr, batch, flags, boxes_dev, scores_dev, cls_dev, count_dev, h->profile != 0);
    h->x_u8 = nullptr;
    return rc;
}

// ---- cv2.resize(image, (W, H)) of BaseTransform (data/__init__.py:36) for uint8 HWC frames, INTER_LINEAR: OpenCV's 8-bit
// fixed-point bilinear (imgproc/resize.cpp: 11-bit coefficients, horizontal pass in int32, vertical pass
// (((b0 * (D0 >> 4)) >> 16) + ((b1 * (D1 >> 4)) >> 16) + 2) >> 2).  The coefficient tables are computed on the host with
// OpenCV's own float / double expressions, so the kernel only gathers and does integer arithmetic.
// Horizontal axis: OpenCV clamps the offset AND zeroes the fraction at both borders.  Vertical axis (`vertical`): it keeps
// floor(f) and the coefficient pair as they are and clips the two ROW INDICES instead (resizeGeneric_Invoker:
// `sy = clip(sy0 - ksize2 + 1 + k, 0, ssize.height)`), so a border row is blended with itself through two separately
// truncated products -- up to 1 LSB below the single-product form; the kernel does the same clipping.
static void linear_tables(int src, int dst, int *ofs, int *coef, bool vertical) {
    const double scale = (double)src / (double)dst;
    for (int d = 0; d < dst; ++d) {
        float f = (float)((d + 0.5) * scale - 0.5);
        int sx = (int)std::floor(f);
        f -= (float)sx;
        if (!vertical && sx < 0) { sx = 0; f = 0.f; }
        if (!vertical && sx >= src - 1) { sx = src - 1; f = 0.f; }
        ofs[d] = sx;
        const long c0 = std::lrintf((1.f - f) * 2048.f), c1 = std::lrintf(f * 2048.f);     // saturate_cast<short>: round half to even
        coef[2 * d] = (int)std::min(32767l, std::max(-32768l, c0));
        coef[2 * d + 1] = (int)std::min(32767l, std::max(-32768l, c1));
    }
}

__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t *src, uint8_t *dst, const int *tab, int sh, int sw, int dh, int dw) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= dh * dw) return;
    const int dy = i / dw, dx = i % dw;
    const int *xofs = tab, *xa = tab + dw, *yofs = tab + 3 * dw, *yb = tab + 3 * dw + dh;
    const int sx0 = xofs[dx], sx1 = min(sx0 + 1, sw - 1), a0 = xa[2 * dx], a1 = xa[2 * dx + 1];
    const int sy0 = min(max(yofs[dy], 0), sh - 1), sy1 = min(max(yofs[dy] + 1, 0), sh - 1), b0 = yb[2 * dy], b1 = yb[2 * dy + 1];
    const uint8_t *s = src + (size_t)b * sh * sw * 3;
    uint8_t *d = dst + ((size_t)b * dh * dw + i) * 3;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int d0 = (int)s[((size_t)sy0 * sw + sx0) * 3 + c] * a0 + (int)s[((size_t)sy0 * sw + sx1) * 3 + c] * a1;
        const int d1 = (int)s[((size_t)sy1 * sw + sx0) * 3 + c] * a0 + (int)s[((size_t)sy1 * sw + sx1) * 3 + c] * a1;
        const int v = (((b0 * (d0 >> 4)) >> 16) + ((b1 * (d1 >> 4)) >> 16) + 2) >> 2;
        d[c] = (uint8_t)min(max(v, 0), 255);
    }
}

// frames of any size: cv2.resize to the network size on the GPU, then y355_forward_u8.  frames_dev uint8 [B][src_h][src_w][3].
// `resized_out_dev` (optional, [B][H][W][3]) receives the resized frames (parity tap).
extern "C" int y355_forward_u8_resized(y355_engine *h, const uint8_t *frames_dev, int src_h, int src_w, int batch, int flags,
                                       float *boxes_dev, float *scores_dev, int32_t *cls_dev, int32_t *count_dev,
                                       uint8_t *resized_out_dev) {
    if (!h || !frames_dev) return fail(Y355_EINVAL, "null argument");
    if (batch < 1 || batch > h->cfg.max_batch) return fail(Y355_EINVAL, "batch out of range");
    if (src_h < 1 || src_w < 1 || src_h > 16384 || src_w > 16384) return fail(Y355_EINVAL, "bad frame size");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    const int H = h->cfg.height, W = h->cfg.width;
    if (!h->rs_frames) {
        if (int rc = dmalloc(h, (void **)&h->rs_frames, (size_t)h->cfg.max_batch * H * W * 3, false)) return rc;
        if (int rc = dmalloc(h, (void **)&h->rs_tab, sizeof(int) * 3 * (size_t)(H + W), false)) return rc;
    }
    if (h->rs_src_h != src_h || h->rs_src_w != src_w) {
        std::vector<int> tab(3 * (size_t)(H + W));
        linear_tables(src_w, W, tab.data(), tab.data() + W, false);
        linear_tables(src_h, H, tab.data() + 3 * W, tab.data() + 3 * W + H, true);
        HIPCHK(hipStreamSynchronize(h->stream));          // a previous forward may still read the old tables
        HIPCHK(hipMemcpy(h->rs_tab, tab.data(), sizeof(int) * tab.size(), hipMemcpyHostToDevice));
        h->rs_src_h = src_h;
        h->rs_src_w = src_w;
    }
    hipLaunchKernelGGL(resize_u8_kernel, dim3((H * W + 255) / 256, batch), dim3(256), 0, h->stream, frames_dev, h->rs_frames, h->rs_tab,
                       src_h, src_w, H, W);
    HIPCHK(hipGetLastError());
    if (resized_out_dev)
        HIPCHK(hipMemcpyAsync(resized_out_dev, h->rs_frames, (size_t)batch * H * W * 3, hipMemcpyDeviceToDevice, h->stream));
    if (!boxes_dev && !scores_dev && !cls_dev && !count_dev) return 0;      // resize only
    return y355_forward_u8(h, h->rs_frames, batch, flags, boxes_dev, scores_dev, cls_dev, count_dev);
}

extern "C" int y355_forward_host(y355_engine *h, const float *x_host, int batch, int flags, float *boxes,
                                 float *scores, int32_t *cls, int32_t *count) {
    if (!h || !x_host || !boxes || !scores || !cls || !count) return fail(Y355_EINVAL, "null argument");
    if (batch < 1 || batch > h->cfg.max_batch) return fail(Y355_EINVAL, "batch out of range");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    const size_t xin = (size_t)3 * h->cfg.height * h->cfg.width;
    if (!h->x_stage) {
        if (int rc = dmalloc(h, (void **)&h->x_stage, sizeof(float) * xin * h->cfg.max_batch, false)) return rc;
    }
    HIPCHK(hipMemcpyAsync(h->x_stage, x_host, sizeof(float) * xin * batch, hipMemcpyHostToDevice, h->stream));
    if (int rc = y355_forward(h, h->x_stage, batch, flags, h->o_box, h->o_score, h->o_cls, h->o_count)) return rc;
    const size_t md = h->max_det;
    HIPCHK(hipMemcpyAsync(boxes, h->o_box, sizeof(float) * 4 * md * batch, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(scores, h->o_score, sizeof(float) * md * batch, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(cls, h->o_cls, sizeof(int) * md * batch, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(count, h->o_count, sizeof(int) * batch, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int y355_forward_counters(y355_engine *h, int64_t *saturated, int64_t *guard) {
    if (!h) return fail(Y355_EINVAL, "null engine");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    Counters c[10];
    HIPCHK(hipMemcpyAsync(c, h->ctr_dev, sizeof c, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    int64_t s = 0, g = 0;
    for (int k = 0; k < 10; ++k) {
        s += (int64_t)c[k].sat + c[k].in_sat;
        g += (int64_t)c[k].guard;
    }
    if (saturated) *saturated = s;
    if (guard) *guard = g;
    return 0;
}

extern "C" int y355_get_candidates(y355_engine *h, int batch, float *boxes, float *scores, int32_t *cls) {
    if (!h || !boxes || !scores || !cls) return fail(Y355_EINVAL, "null argument");
    if (batch < 1 || batch > h->cfg.max_batch) return fail(Y355_EINVAL, "batch out of range");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(boxes, h->cand_box, sizeof(float) * 4 * h->N * batch, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(scores, h->cand_score, sizeof(float) * h->N * batch, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(cls, h->cand_cls, sizeof(int) * h->N * batch, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" int y355_head_nms(y355_engine *h, const int8_t *pred_q, int batch, int sa_pred, float *boxes,
                             float *scores, int32_t *cls, int32_t *count) {
    if (!h || !pred_q || !boxes || !scores || !cls || !count) return fail(Y355_EINVAL, "null argument");
    if (batch < 1 || batch > h->cfg.max_batch) return fail(Y355_EINVAL, "batch out of range");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    const Layer &L = h->L[9];
    const int CS = L.cout_pad, Hs = h->Hs, Ws = h->Ws, PC = L.cout;
    std::vector<int8_t> tmp((size_t)batch * Hs * Ws * CS, 0);
    for (int b = 0; b < batch; ++b)
        for (int c = 0; c < PC; ++c)
            for (int y = 0; y < Hs; ++y)
                for (int x = 0; x < Ws; ++x)
                    tmp[(((size_t)b * Hs + y) * Ws + x) * CS + c] = pred_q[(((size_t)b * PC + c) * Hs + y) * Ws + x];
    HIPCHK(hipStreamSynchronize(h->stream));
    HIPCHK(hipMemcpy(L.out_dev, tmp.data(), tmp.size(), hipMemcpyHostToDevice));
    HeadParams hp = head_params(h, sa_pred, h->o_box, h->o_score, h->o_cls, h->o_count);
    y355_launch_head_nms(hp, batch, h->ws, h->stream, nullptr);
    HIPCHK(hipGetLastError());
    const size_t md = h->max_det;
    HIPCHK(hipMemcpyAsync(boxes, h->o_box, sizeof(float) * 4 * md * batch, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(scores, h->o_score, sizeof(float) * md * batch, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(cls, h->o_cls, sizeof(int) * md * batch, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipMemcpyAsync(count, h->o_count, sizeof(int) * batch, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int y355_max_det(y355_engine *h) { return h ? h->max_det : Y355_EINVAL; }
extern "C" int y355_num_anchors_total(y355_engine *h) { return h ? h->N : Y355_EINVAL; }

// Evaluator-side step right after the path: `bboxes *= [[w, h, w, h]]` of every image
// (test.py:88-90, utils/vocapi_evaluator_mask.py:71-72, utils/cocoapi_evaluator.py:77-85) for a whole batch on
// the GPU, on the engine's stream, in place.  The reference's float32 *= int64 computes the product in float64
// and rounds once to float32; box (24-bit) x size (< 2^24) is exact in float64, so one fp32 multiply is identical.
__global__ void scale_boxes_kernel(float *boxes, const int32_t *count, const float *wh, int max_det) {
    const int b = blockIdx.y;
    const int n = count[b];
    const float w = wh[2 * b], h = wh[2 * b + 1];
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        float4 *p = (float4 *)(boxes + ((size_t)b * max_det + i) * 4);
        float4 v = *p;
        v.x *= w; v.y *= h; v.z *= w; v.w *= h;
        *p = v;
    }
}
extern "C" int y355_scale_boxes(y355_engine *h, float *boxes_dev, const int32_t *count_dev, const float *wh_dev, int batch) {
    if (!h || !boxes_dev || !count_dev || !wh_dev) return fail(Y355_EINVAL, "null argument");
    if (batch < 1 || batch > h->cfg.max_batch) return fail(Y355_EINVAL, "batch out of range");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    const int md = y355_max_det(h);
    hipLaunchKernelGGL(scale_boxes_kernel, dim3((md + 255) / 256, batch), dim3(256), 0, h->stream, boxes_dev, count_dev, wh_dev, md);
    HIPCHK(hipGetLastError());
    return 0;
}

extern "C" int y355_pack_front_weights(const int8_t *q_w1, const int8_t *q_w2, int8_t *dst) {
    if (!dst) return fail(Y355_EINVAL, "null argument");
    y355_pack_front(q_w1, q_w2, dst);
    return 0;
}

// diagnostic: record s_memtime stamps of the production conv kernel of `layer` (-1 = off)
extern "C" int y355_debug_stamps(y355_engine *h, int layer, unsigned long long *out_host, int nwg) {
    if (!h) return fail(Y355_EINVAL, "null engine");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    if (!h->stamps_dev) {
        if (int rc = dmalloc(h, (void **)&h->stamps_dev, 8 * 32 * Y355_STAMP_ROWS, true)) return rc;
    }
    if (out_host) {
        HIPCHK(hipStreamSynchronize(h->stream));
        HIPCHK(hipMemcpy(out_host, h->stamps_dev, 8 * 32 * (size_t)(nwg > Y355_STAMP_ROWS ? Y355_STAMP_ROWS : nwg), hipMemcpyDeviceToHost));
    }
    h->stamp_layer = layer;
    HIPCHK(hipMemset(h->stamps_dev, 0, 8 * 32 * Y355_STAMP_ROWS));
    return 0;
}

extern unsigned long long *y355_nms_stamps_dev;
// diagnostics: s_memtime stamps of the head / pairs / resolve kernels of the last run (Y355_NMS_STAMPS=1)
extern "C" int y355_debug_nms_stamps(unsigned long long *out_host) {
    if (!out_host || !y355_nms_stamps_dev) return fail(Y355_ENOTREADY, "set Y355_NMS_STAMPS=1 before the first forward");
    HIPCHK(hipDeviceSynchronize());
    HIPCHK(hipMemcpy(out_host, y355_nms_stamps_dev, 8 * 8 * 256 * 4, hipMemcpyDeviceToHost));
    return 0;
}

extern "C" void *y355_stream(y355_engine *h) { return h ? (void *)h->stream : nullptr; }

extern "C" int y355_sync(y355_engine *h) {
    if (!h) return fail(Y355_EINVAL, "null engine");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipStreamSynchronize(h->stream));
    return 0;
}

extern "C" int y355_profile(y355_engine *h, int enable) {
    if (!h) return fail(Y355_EINVAL, "null engine");
    h->profile = enable;
    return 0;
}

extern "C" int y355_profile_get(y355_engine *h, float *ms) {
    if (!h || !ms) return fail(Y355_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipEventSynchronize(h->ev[12]));
    for (int i = 0; i < Y355_NUM_TIMERS; ++i) HIPCHK(hipEventElapsedTime(&ms[i], h->ev[i], h->ev[i + 1]));
    return 0;
}

extern "C" int y355_profile_kernel_get(y355_engine *h, float *ms) {
    if (!h || !ms) return fail(Y355_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipEventSynchronize(h->ev[12]));
    for (int k = 0; k < 10; ++k) {
        ms[k] = 0.f;
        if (h->kev_set[k]) HIPCHK(hipEventElapsedTime(&ms[k], h->kev[k][0], h->kev[k][1]));
    }
    return 0;
}

extern "C" int y355_profile_kernels_get(y355_engine *h, float *ms) {
    if (!h || !ms) return fail(Y355_EINVAL, "null argument");
    HIPCHK(hipSetDevice(h->cfg.device_id));
    HIPCHK(hipEventSynchronize(h->ev[12]));
    for (int k = 0; k < Y355_NUM_KERNEL_TIMERS; ++k) {
        ms[k] = 0.f;
        if (h->kev_set[k]) HIPCHK(hipEventElapsedTime(&ms[k], h->kev[k][0], h->kev[k][1]));
    }
    return 0;
}

// ------------------------------------------------------------------------------------------
// Operator-level conv + bias + LeakyReLU WITHOUT requantisation: t' (int64) and F' such that the
// reference's Conv2d_fuse output is exactly t' / 2^F' (utils/modules.py:20-29 on fake-quantized
// operands).  Host pointers, synchronous.
extern "C" int y355_conv3x3_i8_raw(int device_id, const int8_t *q_in, const int8_t *q_w, const int32_t *q_b,
                                   int batch, int cin, int cout, int H, int W, int sa_in, int e_w, int e_b,
                                   int flags, int64_t *out, int32_t *frac_bits) {
    if (!q_in || !q_w || !q_b || !out || !frac_bits) return fail(Y355_EINVAL, "null argument");
    if (batch < 1 || cin < 1 || cin > 256 || cout < 1 || H < 1 || W < 1) return fail(Y355_EINVAL, "bad shape (cin <= 256)");
    if ((flags & Y355_OP_LEAKY) && (flags & Y355_OP_RELU)) return fail(Y355_EINVAL, "LeakyReLU and ReLU are exclusive");
    const int leaky = (flags & Y355_OP_LEAKY) ? 1 : ((flags & Y355_OP_RELU) ? 2 : 0);
    HIPCHK(hipSetDevice(device_id));
    if (int e = prepare_kernels()) return e;
    const int cpad = cin <= 16 ? 16 : cin <= 32 ? 32 : cin <= 64 ? 64 : cin <= 128 ? 128 : 256;
    const int sel = cpad == 16 ? 0 : cpad == 32 ? 1 : cpad == 64 ? 2 : cpad == 128 ? 3 : 4;
    const ConvKernelInfo &ki = *y355_conv_kernel(Y355_K_GEN16 + sel);
    const int cout_pad = (cout + ki.bn - 1) / ki.bn * ki.bn;
    Requant rq{};
    int fb = 0;
    std::vector<int32_t> bt;
    std::vector<long long> bw;
    if (int rc = make_requant(cin, sa_in, e_w, e_b, 0, false, leaky, 10, q_b, cout, cout_pad, &rq, &fb, &bt, &bw)) return rc;
    const size_t in_elems = ((size_t)batch * (H + 2) * (W + 2) + 64) * cpad;
    std::vector<int8_t> xin(in_elems, 0);
    for (int b = 0; b < batch; ++b)
        for (int c = 0; c < cin; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x)
                    xin[(((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * cpad + c] =
                        q_in[(((size_t)b * cin + c) * H + y) * W + x];
    std::vector<int8_t> packed(y355_packed_bytes(ki, cout_pad));
    y355_pack_weights(ki, q_w, cout, cin, cout_pad, packed.data());
    const size_t raw_elems = (size_t)batch * H * W * cout_pad;
    int8_t *d_in = nullptr, *d_w = nullptr;
    int *d_b = nullptr;
    long long *d_bw = nullptr, *d_raw = nullptr;
    Counters *d_c = nullptr;
    auto cleanup = [&]() {
        (void)hipFree(d_in); (void)hipFree(d_w); (void)hipFree(d_b); (void)hipFree(d_bw); (void)hipFree(d_raw); (void)hipFree(d_c);
    };
#define RAWCHK(expr)                                                                        \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            cleanup();                                                                      \
            return fail(Y355_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
        }                                                                                   \
    } while (0)
    RAWCHK(hipMalloc((void **)&d_in, in_elems));
    RAWCHK(hipMalloc((void **)&d_w, packed.size()));
    RAWCHK(hipMalloc((void **)&d_b, sizeof(int) * cout_pad));
    RAWCHK(hipMalloc((void **)&d_bw, sizeof(long long) * cout_pad));
    RAWCHK(hipMalloc((void **)&d_raw, sizeof(long long) * raw_elems));
    RAWCHK(hipMalloc((void **)&d_c, sizeof(Counters)));
    RAWCHK(hipMemcpy(d_in, xin.data(), in_elems, hipMemcpyHostToDevice));
    RAWCHK(hipMemcpy(d_w, packed.data(), packed.size(), hipMemcpyHostToDevice));
    RAWCHK(hipMemcpy(d_b, bt.data(), sizeof(int) * cout_pad, hipMemcpyHostToDevice));
    RAWCHK(hipMemcpy(d_bw, bw.data(), sizeof(long long) * cout_pad, hipMemcpyHostToDevice));
    RAWCHK(hipMemset(d_c, 0, sizeof(Counters)));
    ConvParams p{};
    p.in = d_in; p.out = nullptr; p.w = d_w; p.bias_t = d_b; p.bias_w = d_bw; p.ctr = d_c; p.raw = d_raw;
    p.B = batch; p.H = H; p.W = W; p.cstride = cout_pad; p.out_halo = 0;
    p.tiles_x = (W + ki.tw - 1) / ki.tw; p.tiles_y = (H + ki.th - 1) / ki.th; p.nblk = cout_pad / ki.bn;
    p.rq = rq; p.mode = 1; p.guard = 0;
    ki.launch(p, p.tiles_x * p.tiles_y * p.nblk * batch, 0);
    RAWCHK(hipGetLastError());
    RAWCHK(hipDeviceSynchronize());
    std::vector<long long> o(raw_elems);
    RAWCHK(hipMemcpy(o.data(), d_raw, sizeof(long long) * raw_elems, hipMemcpyDeviceToHost));
    cleanup();
    for (int b = 0; b < batch; ++b)
        for (int c = 0; c < cout; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x)
                    out[(((size_t)b * cout + c) * H + y) * W + x] = o[(((size_t)b * H + y) * W + x) * cout_pad + c];
    *frac_bits = fb;
    return 0;
}

// ------------------------------------------------------------------------------------------
// Operator-level fused layer on caller data (utils/modules.py Conv2d_fuse drop-in, unit tests)
extern "C" int y355_conv3x3_i8_fused(int device_id, const int8_t *q_in, const int8_t *q_w, const int32_t *q_b,
                                     int batch, int cin, int cout, int H, int W, int sa_in, int e_w, int e_b,
                                     int sa_out, int flags, int8_t *out, y355_layer_stats *stats) {
    if (!q_in || !q_w || !q_b || !out) return fail(Y355_EINVAL, "null argument");
    if (batch < 1 || cin < 1 || cin > 256 || cout < 1 || H < 1 || W < 1) return fail(Y355_EINVAL, "bad shape (cin <= 256)");
    if ((flags & Y355_OP_LEAKY) && (flags & Y355_OP_RELU)) return fail(Y355_EINVAL, "LeakyReLU and ReLU are exclusive");
    const int pool = (flags & Y355_OP_POOL) ? 1 : 0, leaky = (flags & Y355_OP_LEAKY) ? 1 : ((flags & Y355_OP_RELU) ? 2 : 0);
    if (pool && ((H | W) & 1)) return fail(Y355_EINVAL, "pooling needs even H, W");
    HIPCHK(hipSetDevice(device_id));
    if (int e = prepare_kernels()) return e;
    const int cpad = cin <= 16 ? 16 : cin <= 32 ? 32 : cin <= 64 ? 64 : cin <= 128 ? 128 : 256;
    const int sel = cpad == 16 ? 0 : cpad == 32 ? 1 : cpad == 64 ? 2 : cpad == 128 ? 3 : 4;
    const int kid = (pool ? Y355_K_GEN16P : Y355_K_GEN16) + sel;
    const ConvKernelInfo &ki = *y355_conv_kernel(kid);
    const int cout_pad = (cout + ki.bn - 1) / ki.bn * ki.bn;
    const int Ho = pool ? H / 2 : H, Wo = pool ? W / 2 : W;
    Requant rq{};
    int fb = 0;
    std::vector<int32_t> bt;
    std::vector<long long> bw;
    if (int rc = make_requant(cin, sa_in, e_w, e_b, sa_out, true, leaky, 10, q_b, cout, cout_pad, &rq, &fb, &bt, &bw)) return rc;
    // host-side layout conversion: NCHW -> NHWC with halo and zero channel padding
    const size_t in_elems = ((size_t)batch * (H + 2) * (W + 2) + 64) * cpad;
    std::vector<int8_t> xin(in_elems, 0);
    for (int b = 0; b < batch; ++b)
        for (int c = 0; c < cin; ++c)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x)
                    xin[(((size_t)b * (H + 2) + y + 1) * (W + 2) + x + 1) * cpad + c] =
                        q_in[(((size_t)b * cin + c) * H + y) * W + x];
    std::vector<int8_t> packed(y355_packed_bytes(ki, cout_pad));
    y355_pack_weights(ki, q_w, cout, cin, cout_pad, packed.data());
    const size_t out_elems = (size_t)batch * Ho * Wo * cout_pad;
    int8_t *d_in = nullptr, *d_w = nullptr, *d_out = nullptr;
    int *d_b = nullptr;
    long long *d_bw = nullptr;
    Counters *d_c = nullptr;
    int rc = 0;
    auto cleanup = [&]() {
        (void)hipFree(d_in); (void)hipFree(d_w); (void)hipFree(d_out); (void)hipFree(d_b); (void)hipFree(d_bw); (void)hipFree(d_c);
    };
#define OPCHK(expr)                                                                         \
    do {                                                                                    \
        hipError_t e_ = (expr);                                                             \
        if (e_ != hipSuccess) {                                                             \
            cleanup();                                                                      \
            return fail(Y355_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));      \
        }                                                                                   \
    } while (0)
    OPCHK(hipMalloc((void **)&d_in, in_elems));
    OPCHK(hipMalloc((void **)&d_w, packed.size()));
    OPCHK(hipMalloc((void **)&d_out, out_elems + 64));
    OPCHK(hipMalloc((void **)&d_b, sizeof(int) * cout_pad));
    OPCHK(hipMalloc((void **)&d_bw, sizeof(long long) * cout_pad));
    OPCHK(hipMalloc((void **)&d_c, sizeof(Counters)));
    OPCHK(hipMemcpy(d_in, xin.data(), in_elems, hipMemcpyHostToDevice));
    OPCHK(hipMemcpy(d_w, packed.data(), packed.size(), hipMemcpyHostToDevice));
    OPCHK(hipMemcpy(d_b, bt.data(), sizeof(int) * cout_pad, hipMemcpyHostToDevice));
    OPCHK(hipMemcpy(d_bw, bw.data(), sizeof(long long) * cout_pad, hipMemcpyHostToDevice));
    OPCHK(hipMemset(d_out, 0, out_elems + 64));
    Counters cz{};
    Counters cs{};
    for (int mode = 1; mode >= 0; --mode) {
        OPCHK(hipMemset(d_c, 0, sizeof(Counters)));
        ConvParams p{};
        p.in = d_in; p.out = d_out; p.w = d_w; p.bias_t = d_b; p.bias_w = d_bw; p.ctr = d_c;
        p.B = batch; p.H = H; p.W = W; p.cstride = cout_pad; p.out_halo = 0;
        p.tiles_x = (W + ki.tw - 1) / ki.tw; p.tiles_y = (H + ki.th - 1) / ki.th; p.nblk = cout_pad / ki.bn;
        p.rq = rq; p.mode = mode; p.guard = 1;
        ki.launch(p, p.tiles_x * p.tiles_y * p.nblk * batch, 0);
        OPCHK(hipGetLastError());
        OPCHK(hipDeviceSynchronize());
        OPCHK(hipMemcpy(mode ? &cs : &cz, d_c, sizeof(Counters), hipMemcpyDeviceToHost));
    }
    std::vector<int8_t> o(out_elems);
    OPCHK(hipMemcpy(o.data(), d_out, out_elems, hipMemcpyDeviceToHost));
    cleanup();
    for (int b = 0; b < batch; ++b)
        for (int c = 0; c < cout; ++c)
            for (int y = 0; y < Ho; ++y)
                for (int x = 0; x < Wo; ++x)
                    out[(((size_t)b * cout + c) * Ho + y) * Wo + x] = o[(((size_t)b * Ho + y) * Wo + x) * cout_pad + c];
    if (stats) {
        stats->absmax_t = (int64_t)cs.absmax;
        stats->frac_bits = fb;
        stats->reserved = 0;
        stats->saturated = (int64_t)cz.sat;
        stats->guard = (int64_t)cz.guard;
    }
    (void)rc;
    return 0;
}

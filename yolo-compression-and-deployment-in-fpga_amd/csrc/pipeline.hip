// yolo355 -- y355_pipeline: the throughput regime of the hot path as a product entry point (include/yolo355.h).
//
// The reference's callers hand the network one batch after another (`net(x)` per image in test.py:84 and in the evaluator loops
// utils/vocapi_evaluator_mask.py:57-82, utils/cocoapi_evaluator.py:70-98).  One engine handle on one stream leaves the GPU to the
// detection head / NMS of batch i (a few latency-bound workgroups) before the convolutions of batch i + 1 may start; with a few
// handles on as many HIP streams the tail of one batch runs beside the convolutions of the next.  Rounds 2-5 measured that
// regime with a scheduler inside bench.py; this file is that scheduler behind the C ABI: `handles` y355_engine objects (each
// with its own non-blocking stream, weights, workspaces), tickets dealt round-robin, one HIP event per ticket, and the
// engines' throughput mode (Y355_OPT_RING_WORKGROUPS) set when more than one handle shares the GPU.
//
// Built on the public engine ABI only (y355_create / y355_forward / ...): results are those of a stand-alone engine, bit for bit
// (tests/test_pipeline.py).
#include "../../include/yolo355.h"
#include "y355_common.h"

#include <cstdio>
#include <string>
#include <vector>

int y355_fail(int code, const std::string &msg);     // engine.hip: sets the message y355_last_error() returns

namespace {
int pfail(int code, const std::string &msg) { return y355_fail(code, msg); }
#define PHIPCHK(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return pfail(Y355_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_));    \
    } while (0)

struct Slot {
    float *boxes = nullptr, *scores = nullptr;      // pipeline-owned outputs of the ticket in this slot (allocated on first use)
    int32_t *cls = nullptr, *count = nullptr;
    // what the ticket's forward wrote to (the caller's buffers or the ones above)
    float *o_boxes = nullptr, *o_scores = nullptr;
    int32_t *o_cls = nullptr, *o_count = nullptr;
    hipEvent_t done = nullptr;
    hipEvent_t released = nullptr;                   // recorded by y355_pipeline_release on the consumer's stream
    bool has_release = false;
    long long ticket = -1;
    int batch = 0;
};
}  // namespace

struct y355_pipeline {
    y355_config cfg{};
    std::vector<y355_engine *> eng;
    std::vector<Slot> slots;                         // depth = 2 x handles: ticket t -> handle t % handles, slot t % depth
    hipEvent_t input_ready = nullptr;
    hipStream_t copy_stream = nullptr;               // y355_pipeline_fetch: device -> host behind the ticket's event, beside later forwards
    long long next = 0;
    int ring_wgs = 0;
    int max_det = 0;
};

extern "C" void y355_pipeline_destroy(y355_pipeline *p) {
    if (!p) return;
    (void)hipSetDevice(p->cfg.device_id);
    for (auto *e : p->eng)
        if (e) (void)y355_sync(e);
    for (auto &s : p->slots) {
        (void)hipFree(s.boxes); (void)hipFree(s.scores); (void)hipFree(s.cls); (void)hipFree(s.count);
        if (s.done) (void)hipEventDestroy(s.done);
        if (s.released) (void)hipEventDestroy(s.released);
    }
    if (p->input_ready) (void)hipEventDestroy(p->input_ready);
    if (p->copy_stream) (void)hipStreamDestroy(p->copy_stream);
    for (auto *e : p->eng) y355_destroy(e);
    delete p;
}

extern "C" int y355_pipeline_create_on(const y355_config *cfg, int handles, int ring_workgroups, void *const *streams,
                                       y355_pipeline **out) {
    if (!cfg || !out) return pfail(Y355_EINVAL, "null argument");
    if (streams && handles < 1) return pfail(Y355_EINVAL, "caller streams: say how many (handles >= 1)");
    if (handles == 0) handles = Y355_PIPE_DEFAULT_HANDLES;      // the measured optimum on one MI355X (profiles/r06_notes.md: 2 .. 6 handles)
    if (handles < 1 || handles > 8) return pfail(Y355_EINVAL, "handles must be 1..8 (0 = default)");
    if (ring_workgroups < 0) ring_workgroups = handles > 1 ? 128 : 0;      // same sweep: 96 / 128 / 160 / 192 workgroups per launch
    if (ring_workgroups > 4096) return pfail(Y355_EINVAL, "workgroups per launch out of range");
    y355_pipeline *p = new y355_pipeline();                    // y355_create checks the configuration, then selects the device
    p->cfg = *cfg;
    p->ring_wgs = handles > 1 ? ring_workgroups : 0; // a handle that has the GPU to itself wants one workgroup per CU
    y355_config c = *cfg;
    for (int i = 0; i < handles; ++i) {
        y355_engine *e = nullptr;
        // the caller's streams (a host runtime that tracks memory per stream, e.g. PyTorch's allocator, must own the streams its
        // tensors are used on and outlive them) or engine-owned non-blocking ones
        c.stream = streams ? streams[i] : nullptr;
        c.own_stream = streams ? 0 : 1;
        int rc = y355_create(&c, &e);
        if (!rc) {
            p->eng.push_back(e);
            rc = y355_set_option(e, Y355_OPT_RING_WORKGROUPS, p->ring_wgs);
        }
        if (rc) {
            std::string keep = y355_last_error();
            y355_pipeline_destroy(p);
            return pfail(rc, keep);
        }
    }
    p->max_det = y355_max_det(p->eng[0]);
    p->slots.resize(2 * (size_t)handles);
    bool ok = hipEventCreateWithFlags(&p->input_ready, hipEventDisableTiming) == hipSuccess &&
              hipStreamCreateWithFlags(&p->copy_stream, hipStreamNonBlocking) == hipSuccess;
    for (auto &s : p->slots)
        ok = ok && hipEventCreateWithFlags(&s.done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&s.released, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        y355_pipeline_destroy(p);
        return pfail(Y355_EHIP, "hipEventCreate failed");
    }
    *out = p;
    return 0;
}

extern "C" int y355_pipeline_create(const y355_config *cfg, int handles, int ring_workgroups, y355_pipeline **out) {
    return y355_pipeline_create_on(cfg, handles, ring_workgroups, nullptr, out);
}

extern "C" int y355_pipeline_handles(y355_pipeline *p) { return p ? (int)p->eng.size() : Y355_EINVAL; }
extern "C" int y355_pipeline_depth(y355_pipeline *p) { return p ? (int)p->slots.size() : Y355_EINVAL; }
extern "C" int y355_pipeline_max_det(y355_pipeline *p) { return p ? p->max_det : Y355_EINVAL; }
extern "C" y355_engine *y355_pipeline_engine(y355_pipeline *p, int i) {
    return (p && i >= 0 && i < (int)p->eng.size()) ? p->eng[i] : nullptr;
}
extern "C" void *y355_pipeline_stream(y355_pipeline *p, long long ticket) {
    if (!p || ticket < 0) return nullptr;
    return y355_stream(p->eng[(size_t)(ticket % (long long)p->eng.size())]);
}

// ---- configuration: the same call on every handle
#define FOR_ALL(call)                             \
    do {                                          \
        if (!p) return pfail(Y355_EINVAL, "null pipeline"); \
        for (auto *e : p->eng)                    \
            if (int rc = (call)) return rc;       \
        return 0;                                 \
    } while (0)

extern "C" int y355_pipeline_load_layer(y355_pipeline *p, int idx, const int8_t *q_w, const int32_t *q_b, int cout, int cin,
                                        int e_w, int e_b) {
    FOR_ALL(y355_load_layer(e, idx, q_w, q_b, cout, cin, e_w, e_b));
}
extern "C" int y355_pipeline_set_act_exponents(y355_pipeline *p, const int32_t *sa) { FOR_ALL(y355_set_act_exponents(e, sa)); }
extern "C" int y355_pipeline_set_retune(y355_pipeline *p, const int32_t *retune) { FOR_ALL(y355_set_retune(e, retune)); }
extern "C" int y355_pipeline_set_thresholds(y355_pipeline *p, float conf, float nms) { FOR_ALL(y355_set_thresholds(e, conf, nms)); }
extern "C" int y355_pipeline_set_normalization(y355_pipeline *p, const float *mean_bgr, const float *std_bgr) {
    FOR_ALL(y355_set_normalization(e, mean_bgr, std_bgr));
}
extern "C" int y355_pipeline_set_option(y355_pipeline *p, int option, int value) {
    if (p && option == Y355_OPT_RING_WORKGROUPS) p->ring_wgs = value;
    FOR_ALL(y355_set_option(e, option, value));
}

// calibrate on handle 0 (AveragedRangeTracker semantics, y355_calibrate), then give every handle the same trackers and exponents
extern "C" int y355_pipeline_calibrate(y355_pipeline *p, const float *x_dev, int batch, int freeze, double momentum,
                                       int32_t *sa_out, float *max_out) {
    if (!p) return pfail(Y355_EINVAL, "null pipeline");
    int32_t sa[Y355_NUM_TRACKERS];
    if (int rc = y355_calibrate(p->eng[0], x_dev, batch, freeze, momentum, sa, max_out)) return rc;
    float scale[Y355_NUM_TRACKERS];
    int32_t first[Y355_NUM_TRACKERS];
    if (int rc = y355_get_trackers(p->eng[0], scale, first)) return rc;
    for (size_t i = 1; i < p->eng.size(); ++i) {
        if (int rc = y355_set_trackers(p->eng[i], scale, first)) return rc;
        if (int rc = y355_set_act_exponents(p->eng[i], sa)) return rc;
    }
    if (int rc = y355_sync(p->eng[0])) return rc;
    if (sa_out)
        for (int i = 0; i < Y355_NUM_TRACKERS; ++i) sa_out[i] = sa[i];
    return 0;
}
extern "C" int y355_pipeline_set_trackers(y355_pipeline *p, const float *scale, const int32_t *first_a) {
    FOR_ALL(y355_set_trackers(e, scale, first_a));
}
extern "C" int y355_pipeline_get_trackers(y355_pipeline *p, float *scale, int32_t *first_a) {
    if (!p) return pfail(Y355_EINVAL, "null pipeline");
    return y355_get_trackers(p->eng[0], scale, first_a);
}

// ---- the hot path
static int slot_outputs(y355_pipeline *p, Slot &s) {
    if (s.boxes) return 0;
    const size_t B = (size_t)p->cfg.max_batch, md = (size_t)p->max_det;
    PHIPCHK(hipMalloc((void **)&s.boxes, sizeof(float) * 4 * md * B));
    PHIPCHK(hipMalloc((void **)&s.scores, sizeof(float) * md * B));
    PHIPCHK(hipMalloc((void **)&s.cls, sizeof(int32_t) * md * B));
    PHIPCHK(hipMalloc((void **)&s.count, sizeof(int32_t) * B));
    return 0;
}

static int submit_common(y355_pipeline *p, const void *in_dev, bool u8, int batch, int flags, void *caller_stream, float *boxes_dev,
                         float *scores_dev, int32_t *cls_dev, int32_t *count_dev, long long *ticket) {
    if (!p || !in_dev || !ticket) return pfail(Y355_EINVAL, "null argument");
    const bool own = !boxes_dev && !scores_dev && !cls_dev && !count_dev;
    if (!own && (!boxes_dev || !scores_dev || !cls_dev || !count_dev))
        return pfail(Y355_EINVAL, "give all four output pointers or none (none = pipeline-owned buffers, y355_pipeline_outputs)");
    PHIPCHK(hipSetDevice(p->cfg.device_id));
    const long long t = p->next;
    y355_engine *e = p->eng[(size_t)(t % (long long)p->eng.size())];
    Slot &s = p->slots[(size_t)(t % (long long)p->slots.size())];
    hipStream_t es = (hipStream_t)y355_stream(e);
    if (flags & Y355_PIPE_AFTER_STREAM) {             // the input is produced on the caller's stream: start after what is queued there
        PHIPCHK(hipEventRecord(p->input_ready, (hipStream_t)caller_stream));
        PHIPCHK(hipStreamWaitEvent(es, p->input_ready, 0));
    }
    if (s.has_release) {                              // the consumer of the ticket that used this slot said when it is done with it
        PHIPCHK(hipStreamWaitEvent(es, s.released, 0));
        s.has_release = false;
    }
    if (own) {
        if (int rc = slot_outputs(p, s)) return rc;
        boxes_dev = s.boxes; scores_dev = s.scores; cls_dev = s.cls; count_dev = s.count;
    }
    const int ef = flags & (Y355_F_GUARD | Y355_F_TAP);
    const int rc = u8 ? y355_forward_u8(e, (const uint8_t *)in_dev, batch, ef, boxes_dev, scores_dev, cls_dev, count_dev)
                      : y355_forward(e, (const float *)in_dev, batch, ef, boxes_dev, scores_dev, cls_dev, count_dev);
    if (rc) return rc;
    PHIPCHK(hipEventRecord(s.done, es));
    s.o_boxes = boxes_dev; s.o_scores = scores_dev; s.o_cls = cls_dev; s.o_count = count_dev;
    s.ticket = t;
    s.batch = batch;
    p->next = t + 1;
    *ticket = t;
    return 0;
}

extern "C" int y355_pipeline_submit(y355_pipeline *p, const float *x_dev, int batch, int flags, void *caller_stream,
                                    float *boxes_dev, float *scores_dev, int32_t *cls_dev, int32_t *count_dev, long long *ticket) {
    return submit_common(p, x_dev, false, batch, flags, caller_stream, boxes_dev, scores_dev, cls_dev, count_dev, ticket);
}
extern "C" int y355_pipeline_submit_u8(y355_pipeline *p, const uint8_t *frames_dev, int batch, int flags, void *caller_stream,
                                       float *boxes_dev, float *scores_dev, int32_t *cls_dev, int32_t *count_dev, long long *ticket) {
    return submit_common(p, frames_dev, true, batch, flags, caller_stream, boxes_dev, scores_dev, cls_dev, count_dev, ticket);
}

static int find_slot(y355_pipeline *p, long long ticket, Slot **out) {
    if (!p) return pfail(Y355_EINVAL, "null pipeline");
    if (ticket < 0 || ticket >= p->next) return pfail(Y355_EINVAL, "no such ticket");
    Slot &s = p->slots[(size_t)(ticket % (long long)p->slots.size())];
    if (s.ticket != ticket) {
        char buf[160];
        snprintf(buf, sizeof buf, "ticket %lld is gone: its slot was reused by ticket %lld (a ticket lives for %d more submits)",
                 ticket, s.ticket, (int)p->slots.size());
        return pfail(Y355_ENOTREADY, buf);
    }
    *out = &s;
    return 0;
}

extern "C" int y355_pipeline_wait(y355_pipeline *p, long long ticket, int on_stream, void *caller_stream) {
    Slot *s = nullptr;
    if (int rc = find_slot(p, ticket, &s)) return rc;
    PHIPCHK(hipSetDevice(p->cfg.device_id));
    if (on_stream) PHIPCHK(hipStreamWaitEvent((hipStream_t)caller_stream, s->done, 0));
    else PHIPCHK(hipEventSynchronize(s->done));
    return 0;
}

extern "C" int y355_pipeline_outputs(y355_pipeline *p, long long ticket, float **boxes_dev, float **scores_dev, int32_t **cls_dev,
                                     int32_t **count_dev, int *batch) {
    Slot *s = nullptr;
    if (int rc = find_slot(p, ticket, &s)) return rc;
    if (boxes_dev) *boxes_dev = s->o_boxes;
    if (scores_dev) *scores_dev = s->o_scores;
    if (cls_dev) *cls_dev = s->o_cls;
    if (count_dev) *count_dev = s->o_count;
    if (batch) *batch = s->batch;
    return 0;
}

extern "C" int y355_pipeline_release(y355_pipeline *p, long long ticket, void *caller_stream) {
    Slot *s = nullptr;
    if (int rc = find_slot(p, ticket, &s)) return rc;
    PHIPCHK(hipSetDevice(p->cfg.device_id));
    PHIPCHK(hipEventRecord(s->released, (hipStream_t)caller_stream));
    s->has_release = true;
    return 0;
}

// wait for the ticket and copy its detections to host arrays shaped like y355_forward_host's; synchronous
extern "C" int y355_pipeline_fetch(y355_pipeline *p, long long ticket, float *boxes, float *scores, int32_t *cls, int32_t *count) {
    Slot *s = nullptr;
    if (int rc = find_slot(p, ticket, &s)) return rc;
    if (!boxes || !scores || !cls || !count) return pfail(Y355_EINVAL, "null argument");
    PHIPCHK(hipSetDevice(p->cfg.device_id));
    hipStream_t es = p->copy_stream;
    PHIPCHK(hipStreamWaitEvent(es, s->done, 0));
    const size_t md = (size_t)p->max_det, B = (size_t)s->batch;
    PHIPCHK(hipMemcpyAsync(boxes, s->o_boxes, sizeof(float) * 4 * md * B, hipMemcpyDeviceToHost, es));
    PHIPCHK(hipMemcpyAsync(scores, s->o_scores, sizeof(float) * md * B, hipMemcpyDeviceToHost, es));
    PHIPCHK(hipMemcpyAsync(cls, s->o_cls, sizeof(int32_t) * md * B, hipMemcpyDeviceToHost, es));
    PHIPCHK(hipMemcpyAsync(count, s->o_count, sizeof(int32_t) * B, hipMemcpyDeviceToHost, es));
    PHIPCHK(hipStreamSynchronize(es));
    return 0;
}

// the evaluators' `bboxes *= [[w, h, w, h]]` for the ticket's batch (y355_scale_boxes) on the ticket's own stream, in place;
// call it right after the submit (before the next ticket of the same handle); the ticket's event moves behind it
extern "C" int y355_pipeline_scale_boxes(y355_pipeline *p, long long ticket, const float *wh_dev) {
    Slot *s = nullptr;
    if (int rc = find_slot(p, ticket, &s)) return rc;
    y355_engine *e = p->eng[(size_t)(ticket % (long long)p->eng.size())];
    if (int rc = y355_scale_boxes(e, s->o_boxes, s->o_count, wh_dev, s->batch)) return rc;
    PHIPCHK(hipEventRecord(s->done, (hipStream_t)y355_stream(e)));
    return 0;
}

// sums over layers and handles of the saturation / guard counters of each handle's LAST forward; synchronous
extern "C" int y355_pipeline_counters(y355_pipeline *p, int64_t *saturated, int64_t *guard) {
    if (!p) return pfail(Y355_EINVAL, "null pipeline");
    int64_t st = 0, gt = 0;
    const long long n = p->next < (long long)p->eng.size() ? p->next : (long long)p->eng.size();
    for (long long i = 0; i < n; ++i) {
        int64_t s1 = 0, g1 = 0;
        if (int rc = y355_forward_counters(p->eng[(size_t)i], &s1, &g1)) return rc;
        st += s1;
        gt += g1;
    }
    if (saturated) *saturated = st;
    if (guard) *guard = gt;
    return 0;
}

extern "C" int y355_pipeline_sync(y355_pipeline *p) { FOR_ALL(y355_sync(e)); }

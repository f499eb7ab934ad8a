// yolo355 -- multi-GPU exchange of the path: the batch is sharded over the GPUs of one node, one process
// per GPU, no collective on the data path; the only exchange is ONE RCCL all-gather per batch of the fixed-cap padded
// detections, packed into a single buffer.  The reference has no multi-GPU code at all: this replaces
// nothing there, it is the exchange step north_star asks for.
//
// Record layout (one per image, y355_packed_det_bytes(max_det) = 16 + 24 max_det rounded up to 16 bytes):
//   i32 count (-1 = padding record of a ragged shard), i32 pad[3], f32 boxes[max_det][4], f32 scores[max_det],
//   i32 cls[max_det]; entries at or past `count` (and the rounding pad) are zero, so equal detections give equal bytes.
// RCCL is bound at run time (dlopen): the copy already in the process (PyTorch bundles one) is reused when there is one,
// so a process never holds two RCCL runtimes.
#include "../../include/yolo355.h"
#include "y355_common.h"

#include <dlfcn.h>
#include <cstring>
#include <string>

int y355_fail(int code, const std::string &msg);
#define COMMCHK(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return y355_fail(Y355_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
    } while (0)

namespace {
// the slice of the NCCL API this file uses (rccl.h: ncclUniqueId is 128 bytes, ncclChar = 0)
typedef struct { char internal[128]; } nccl_uid;
typedef void *nccl_comm;
struct Rccl {
    void *lib = nullptr;
    int (*GetUniqueId)(nccl_uid *) = nullptr;
    int (*CommInitRank)(nccl_comm *, int, nccl_uid, int) = nullptr;
    int (*CommDestroy)(nccl_comm) = nullptr;
    int (*AllGather)(const void *, void *, size_t, int, nccl_comm, hipStream_t) = nullptr;
    const char *(*GetErrorString)(int) = nullptr;
    std::string err;
};
Rccl g_rccl;

int load_rccl() {
    if (g_rccl.lib) return 0;
    const char *names[] = {"librccl.so", "librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)                       // a copy that is already mapped (torch's) first
        if (!h) h = dlopen(n, RTLD_NOLOAD | RTLD_LAZY | RTLD_GLOBAL);
    for (const char *n : names)
        if (!h) h = dlopen(n, RTLD_LAZY | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_LAZY | RTLD_GLOBAL);
    if (!h) {
        const char *de = dlerror();                   // one call: dlerror() clears the message it returns
        return y355_fail(Y355_ENOTREADY, std::string("RCCL is not loadable: ") + (de ? de : "?"));
    }
    g_rccl.GetUniqueId = (int (*)(nccl_uid *))dlsym(h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (int (*)(nccl_comm *, int, nccl_uid, int))dlsym(h, "ncclCommInitRank");
    g_rccl.CommDestroy = (int (*)(nccl_comm))dlsym(h, "ncclCommDestroy");
    g_rccl.AllGather = (int (*)(const void *, void *, size_t, int, nccl_comm, hipStream_t))dlsym(h, "ncclAllGather");
    g_rccl.GetErrorString = (const char *(*)(int))dlsym(h, "ncclGetErrorString");
    if (!g_rccl.GetUniqueId || !g_rccl.CommInitRank || !g_rccl.CommDestroy || !g_rccl.AllGather)
        return y355_fail(Y355_ENOTREADY, "RCCL library lacks ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather");
    g_rccl.lib = h;
    return 0;
}
int nccl_fail(const char *what, int rc) {
    return y355_fail(Y355_EHIP, std::string(what) + ": " + (g_rccl.GetErrorString ? g_rccl.GetErrorString(rc) : "RCCL error"));
}
}  // namespace

struct y355_comm {
    nccl_comm comm = nullptr;
    int world = 1, rank = 0, device_id = 0;
};

__host__ __device__ static inline size_t rec_bytes_of(int max_det) { return (size_t)16 + ((size_t)24 * (size_t)max_det + 15) / 16 * 16; }
extern "C" size_t y355_packed_det_bytes(int max_det) { return max_det < 0 ? 0 : rec_bytes_of(max_det); }

extern "C" int y355_comm_unique_id(void *id_out) {
    if (!id_out) return y355_fail(Y355_EINVAL, "null argument");
    if (int rc = load_rccl()) return rc;
    nccl_uid id;
    if (int rc = g_rccl.GetUniqueId(&id)) return nccl_fail("ncclGetUniqueId", rc);
    memcpy(id_out, &id, sizeof id);
    return 0;
}

extern "C" int y355_comm_init(y355_comm **out, int world, int rank, const void *id, int device_id) {
    if (!out || !id) return y355_fail(Y355_EINVAL, "null argument");
    if (world < 1 || rank < 0 || rank >= world) return y355_fail(Y355_EINVAL, "bad world / rank");
    if (int rc = load_rccl()) return rc;
    COMMCHK(hipSetDevice(device_id));
    nccl_uid uid;
    memcpy(&uid, id, sizeof uid);
    y355_comm *c = new y355_comm();
    c->world = world;
    c->rank = rank;
    c->device_id = device_id;
    if (int rc = g_rccl.CommInitRank(&c->comm, world, uid, rank)) {
        delete c;
        return nccl_fail("ncclCommInitRank", rc);
    }
    *out = c;
    return 0;
}

extern "C" void y355_comm_destroy(y355_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device_id);
    if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

extern "C" int y355_comm_world(y355_comm *c) { return c ? c->world : Y355_EINVAL; }
extern "C" int y355_comm_rank(y355_comm *c) { return c ? c->rank : Y355_EINVAL; }

// one thread per 16 bytes of the record; grid (ceil(rec16 / 256), records)
// `src_md`: detections per image the source arrays are laid out for (>= max_det, the record's cap: longer lists are cut there)
__global__ __launch_bounds__(256) void pack_dets_kernel(const float4 *boxes, const float *scores, const int *cls, const int *count,
                                                        int batch, int src_md, int max_det, v4i *packed) {
    const int b = blockIdx.y;
    const int rec16 = 1 + max_det + (max_det * 2 + 3) / 4;            // 16-byte units: header, boxes, scores + cls (rounded up)
    const int u = blockIdx.x * 256 + threadIdx.x;
    if (u >= rec16) return;
    const size_t rec_bytes = rec_bytes_of(max_det);
    char *rec = (char *)packed + (size_t)b * rec_bytes;
    const int full = b < batch ? max(count[b], 0) : 0;
    const int n = min(full, max_det);
    if (u == 0) {
        // header: detections kept, detections the image had (> kept: the record was cut at max_det), 0, 0
        *(v4i *)rec = (v4i){b < batch ? n : -1, b < batch ? full : -1, 0, 0};
    } else if (u <= max_det) {
        const int i = u - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < n) v = boxes[(size_t)b * src_md + i];
        *(float4 *)(rec + 16 + (size_t)i * 16) = v;
    } else {
        // scores [max_det] then cls [max_det], 4 dwords per thread (the tail of the record may be shorter)
        const int d0 = (u - 1 - max_det) * 4;
        int *dst = (int *)(rec + 16 + (size_t)max_det * 16);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int d = d0 + k;
            if (d >= (2 * max_det + 3) / 4 * 4) break;   // includes the zeroed rounding pad of the record
            int v = 0;
            if (d < max_det) { if (d < n) v = __float_as_int(scores[(size_t)b * src_md + d]); }
            else if (d < 2 * max_det) { if (d - max_det < n) v = cls[(size_t)b * src_md + d - max_det]; }
            dst[d] = v;
        }
    }
}

extern "C" int y355_pack_dets_capped(const float *boxes_dev, const float *scores_dev, const int32_t *cls_dev, const int32_t *count_dev,
                                     int batch, int records, int src_max_det, int max_det, void *packed_dev, void *stream) {
    if (!boxes_dev || !scores_dev || !cls_dev || !count_dev || !packed_dev) return y355_fail(Y355_EINVAL, "null argument");
    if (batch < 0 || records < batch || records < 1 || max_det < 1 || src_max_det < max_det)
        return y355_fail(Y355_EINVAL, "bad batch / records / max_det");
    const int rec16 = 1 + max_det + (max_det * 2 + 3) / 4;
    hipLaunchKernelGGL(pack_dets_kernel, dim3((rec16 + 255) / 256, records), dim3(256), 0, (hipStream_t)stream,
                       (const float4 *)boxes_dev, scores_dev, cls_dev, count_dev, batch, src_max_det, max_det, (v4i *)packed_dev);
    COMMCHK(hipGetLastError());
    return 0;
}

extern "C" int y355_pack_dets(const float *boxes_dev, const float *scores_dev, const int32_t *cls_dev, const int32_t *count_dev,
                              int batch, int records, int max_det, void *packed_dev, void *stream) {
    return y355_pack_dets_capped(boxes_dev, scores_dev, cls_dev, count_dev, batch, records, max_det, max_det, packed_dev, stream);
}

extern "C" int y355_allgather_dets(y355_comm *c, const void *packed_send_dev, void *packed_recv_dev, int records, int max_det,
                                   void *stream) {
    if (!c || !packed_send_dev || !packed_recv_dev) return y355_fail(Y355_EINVAL, "null argument");
    if (records < 1 || max_det < 1) return y355_fail(Y355_EINVAL, "bad records / max_det");
    COMMCHK(hipSetDevice(c->device_id));
    const size_t bytes = (size_t)records * y355_packed_det_bytes(max_det);
    if (int rc = g_rccl.AllGather(packed_send_dev, packed_recv_dev, bytes, /*ncclChar*/ 0, c->comm, (hipStream_t)stream))
        return nccl_fail("ncclAllGather", rc);
    return 0;
}

// gathered records -> padded arrays in record order; padding records (count -1) are skipped, so the outputs hold the
// images in global order when rank r's shard is images [lo_r, hi_r) (y355 shards are contiguous).  *nimages_out (host)
// is not produced here: callers know the global batch; `out_count` gets one entry per kept record.
__global__ __launch_bounds__(256) void unpack_dets_kernel(const char *packed, const int *slot, int nrec, int max_det, float4 *boxes,
                                                          float *scores, int *cls, int *count) {
    const int r = blockIdx.y;
    const int dst = slot[r];
    if (dst < 0) return;
    const size_t rec_bytes = rec_bytes_of(max_det);
    const char *rec = packed + (size_t)r * rec_bytes;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i == 0) count[dst] = *(const int *)rec;
    if (i < max_det) {
        boxes[(size_t)dst * max_det + i] = *(const float4 *)(rec + 16 + (size_t)i * 16);
        const int *sc = (const int *)(rec + 16 + (size_t)max_det * 16);
        scores[(size_t)dst * max_det + i] = __int_as_float(sc[i]);
        cls[(size_t)dst * max_det + i] = sc[max_det + i];
    }
}

extern "C" int y355_unpack_dets(const void *packed_dev, const int32_t *slot_dev, int records, int max_det, float *boxes_dev,
                                float *scores_dev, int32_t *cls_dev, int32_t *count_dev, void *stream) {
    if (!packed_dev || !slot_dev || !boxes_dev || !scores_dev || !cls_dev || !count_dev) return y355_fail(Y355_EINVAL, "null argument");
    if (records < 1 || max_det < 1) return y355_fail(Y355_EINVAL, "bad records / max_det");
    hipLaunchKernelGGL(unpack_dets_kernel, dim3((max_det + 255) / 256, records), dim3(256), 0, (hipStream_t)stream,
                       (const char *)packed_dev, slot_dev, records, max_det, (float4 *)boxes_dev, scores_dev, cls_dev, count_dev);
    COMMCHK(hipGetLastError());
    return 0;
}

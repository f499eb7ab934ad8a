// yolo355 -- detection head on the GPU: decode + score + threshold + per-class greedy NMS,
// batched (the reference post-processes image 0 only, on the CPU, in Python).
//
// Replaces models/slim_yolo_v2.py:330-358:
//   head split :330-341   (channel layout [A obj | A*C cls | A*4 txtytwth], anchor-major)
//   decode     :111-143   cx=(sig(tx)+gx)*16, w=exp(tw)*aw*16, x1y1x2y2, /[w,h,w,h], clamp
//   score      :348-350   sigmoid(obj) * softmax(cls)
//   postprocess:176-210   argmax class, score >= conf_thresh, per-class NMS, anchor order
//   nms        :145-174   greedy by descending score; suppressed unless iou <= nms_thresh
// and the C head of c_embedding/yolo_forward.c:965-1147 (get_boxes / conf_sort / NMS).
//
// Tie order (undefined in the reference: unstable argsort): (score desc, anchor index asc).
//
// Greedy NMS is a walk over the "i suppresses j" relation in score order.  The relation is
// sparse (a box only interacts with boxes of similar size whose centre is close), so:
//   decode_kernel one thread per anchor: grid/anchor decode, sigmoid / softmax score, best class;
//   head_kernel  one workgroup per image: threshold and counting-sort the candidates
//                into (area octave, centre-bin) order -- bins are a <= 16 x 16 grid over the CLAMPED
//                box centres -- plus per-group extents (max w/h, min/max area);
//   pairs_kernel one thread per candidate walks only the bins a suppressor can sit in
//                (|dcx| < (1-thr)(wi+wj)/2, same in y, area ratio > thr), each unordered pair
//                once, and appends the suppressing pairs to the image's edge list;
//   resolve_emit_kernel one workgroup per image: the endpoints of the edges are settled by rounds
//                ("an earlier kept neighbour kills, an earlier undecided one blocks") over the edge
//                list in LDS; survivors go out in anchor-index order into the padded outputs.
// The pruning is exact: a pair is skipped only when real IoU < 0.999*thr and the union is not
// degenerate, where the fp32 formula of the reference cannot exceed thr.
#include "y355_common.h"
#include <algorithm>
#include <cstdlib>

#define NMS_CAP Y355_NMS_CAP   // max anchors per image handled by this head (416x416: 3380)
#define NBLK (NMS_CAP / 64)
#define MAXA Y355_HEAD_MAXA
#define MAXG Y355_HEAD_MAXG
#define NGROUP 16                 // candidate groups (area octaves) of the sort; NGROUP * Hb * Wb <= NMS_CAP
#define EDGE_CAP Y355_HEAD_EDGE_CAP   // suppressing pairs per image the global list holds (= what resolve_emit_kernel's LDS lists hold:
                                   // pairs_kernel stops writing an image's list beyond that and raises the overflow flag)
#define WG_EDGE_CAP 8192         // edges one pairs workgroup buffers in LDS

struct HeadWork {
    float *cbox;          // [B][CAP][4]  compacted candidates, (anchor, bin) order
    float *cscore;        // [B][CAP]
    int *ccls;            // [B][CAP]
    int *corig;           // [B][CAP]     anchor index n = cell*A + a of compact position p
    int *count;           // [B]          candidates per image
    unsigned int *edges;  // [B][EDGE_CAP] suppressing pairs (p << 12) | q with p < q (compact positions)
    int *nedges;          // [B][2]        number of edges; overflow flag (a list did not fit)
    int *binstart;        // [B][CAP+8]   first compact position of bin (a*HW + by*Ws + bx)
    float *astat;         // [B][MAXG][4] per candidate group: wmax, hmax, amin, amax (clamped boxes)
    int *tiny;            // [B][CAP]     positions of candidates with area < AREA_MIN
    int *ntiny;           // [B]
    int *ctype;           // [B][CAP]     candidate group of compact position p
    float *dbox;          // [B][CAP][4]  decode of every anchor, (level, anchor, cell) order
    float *dscore;        // [B][CAP]
    int *dcls;            // [B][CAP]
    // heads with more than CAP anchors per image (three-level models at 416 x 416): decode_kernel writes the raw arrays
    // (pitch rstride), compact_kernel keeps the anchors at or above conf_thresh in anchor-index order in dbox / dscore /
    // dcls (at most CAP of them; more sets ovf[b]); the sort then sees an ordinary <= CAP-anchor image
    float *rbox;          // [B][rstride][4] or null (= small head: decode writes dbox directly)
    float *rscore;        // [B][rstride]
    int *rcls;            // [B][rstride]
    int *rcount;          // [B] anchors kept by the compaction
    int *ovf;             // [B] 1: more than CAP anchors passed the threshold (the rest were dropped)
    int rstride;
    unsigned long long *stamps;   // diagnostics or null
};

// diagnostics (Y355_NMS_STAMPS=1): s_memtime at the phase boundaries of the first 256 workgroups
#define NSTAMP(k, wg, slot)                                                                        \
    do {                                                                                          \
        if (wk.stamps && threadIdx.x == 0 && (wg) < 256)                                          \
            wk.stamps[(((k) * 256 + (wg)) * 8) + (slot)] = __builtin_amdgcn_s_memtime();          \
    } while (0)
#define PRUNE_MARGIN 1.001f
#define PRUNE_EPS 1e-6f
#define AREA_MIN 1e-10f

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// ---- decode_kernel: one anchor per thread, output in the (level, anchor, cell) enumeration the sort below consumes.
// The decode is ~500 VALU instructions per anchor (9 expf, ~14 IEEE divisions): spread over every CU instead of the 64
// workgroups of the per-image sort.  A workgroup owns DEC_CELLS(A) consecutive cells of one level and all their anchors:
// the cells' prediction vectors (one contiguous range of the NHWC map) are staged in LDS with 16-byte loads, and every
// anchor's channel walk reads LDS.  (Round 2 read the map straight from global memory, consecutive lanes a whole pixel
// apart: byte / dword loads that each touched 64 cache lines -- 860 MB fetched per launch on the YOLOv3tiny graph for
// 14 MB of maps, 148 us)
#define DEC_THREADS 256
__host__ __device__ static inline int dec_cells(int A) { return DEC_THREADS / A; }
__global__ __launch_bounds__(DEC_THREADS) void decode_kernel(const HeadParams p, const HeadWork wk, const int pitch, const int ncell_wg) {
    extern __shared__ __attribute__((aligned(16))) char dsm[];        // dec_cells(A) x pitch bytes (pitch = cell bytes + 4: no bank conflicts)
    const int b = blockIdx.y;
    const int A = p.A, C = p.C;
    const int NCELL = ncell_wg;               // cells per workgroup: dec_cells(A), fewer when their staged vectors would not fit the LDS
    const int HW0 = p.lev[0].Hs * p.lev[0].Ws, N0 = HW0 * A;
    const int HW1 = p.nlev > 1 ? p.lev[1].Hs * p.lev[1].Ws : 0, N1 = N0 + HW1 * A;
    const int HW2 = p.nlev > 2 ? p.lev[2].Hs * p.lev[2].Ws : 0;
    const int N = N1 + HW2 * A;
    const int nb0 = (HW0 + NCELL - 1) / NCELL, nb1 = (HW1 + NCELL - 1) / NCELL;
    int blk = blockIdx.x;
    const int lv = (blk >= nb0 ? 1 : 0) + (blk >= nb0 + nb1 ? 1 : 0);
    blk -= lv == 0 ? 0 : (lv == 1 ? nb0 : nb0 + nb1);
    const HeadLevel &L = p.lev[lv];
    const int lbase = lv == 0 ? 0 : (lv == 1 ? N0 : N1);
    const int HWl = lv == 0 ? HW0 : (lv == 1 ? HW1 : HW2);
    const int cell0 = blk * NCELL, ncell = min(NCELL, HWl - cell0);
    const int es = L.pred ? 1 : 4, cellb = L.cstride * es;            // bytes per cell (a multiple of 16: cstride is padded)
    {
        const char *src = (L.pred ? (const char *)L.pred : (const char *)L.pred_f) + ((size_t)b * HWl + cell0) * cellb;
        const int n16 = ncell * cellb / 16, per = cellb / 16;
        for (int k = threadIdx.x; k < n16; k += DEC_THREADS) {
            const v4i v = *(const v4i *)(src + (size_t)k * 16);
            const int c = k / per, w = k - c * per;
            int *dst = (int *)(dsm + c * pitch + w * 16);
            dst[0] = v[0]; dst[1] = v[1]; dst[2] = v[2]; dst[3] = v[3];
        }
    }
    __syncthreads();
    const int a = threadIdx.x / NCELL, lc = threadIdx.x - a * NCELL;
    if (a >= A || lc >= ncell) return;
    const int cell = cell0 + lc;
    const int np = lbase + a * HWl + cell;      // slot in the (level, anchor, cell) order
    const int n = lbase + cell * A + a;        // the reference's anchor index (:337-341)
    const int gy = cell / L.Ws, gx = cell % L.Ws;
    const char *row = dsm + lc * pitch;
    const bool i8 = L.pred != nullptr;
    const float dq = L.dq;
    auto ld = [&](int c) -> float { return i8 ? (float)((const int8_t *)row)[c] * dq : ((const float *)row)[c]; };
    const float conf = ld(a);
    const float obj = sigmoidf_(conf);
    const int c0 = A + a * C;
    float m = -3.0e38f;
    for (int c = 0; c < C; ++c) m = fmaxf(m, ld(c0 + c));
    float sum = 0.f;
    for (int c = 0; c < C; ++c) sum += expf(ld(c0 + c) - m);
    float best = -1.f;
    int bc = 0;
    for (int c = 0; c < C; ++c) {
        const float s = (expf(ld(c0 + c) - m) / sum) * obj;
        if (s > best) { best = s; bc = c; }
    }
    const int t0 = A * (1 + C) + a * 4;
    const float tx = ld(t0), tyy = ld(t0 + 1), tw = ld(t0 + 2), th = ld(t0 + 3);
    const float cx = (sigmoidf_(tx) + (float)gx) * L.stride;
    const float cy = (sigmoidf_(tyy) + (float)gy) * L.stride;
    const float bw = (expf(tw) * L.anchors[2 * a]) * p.wh_mul;
    const float bh = (expf(th) * L.anchors[2 * a + 1]) * p.wh_mul;
    float4 bx4;
    bx4.x = fminf(fmaxf((cx - bw / 2) / p.in_w, 0.f), 1.f);
    bx4.y = fminf(fmaxf((cy - bh / 2) / p.in_h, 0.f), 1.f);
    bx4.z = fminf(fmaxf((cx + bw / 2) / p.in_w, 0.f), 1.f);
    bx4.w = fminf(fmaxf((cy + bh / 2) / p.in_h, 0.f), 1.f);
    if (wk.rbox) {
        const size_t o = (size_t)b * wk.rstride + np;
        ((float4 *)wk.rbox)[o] = bx4;
        wk.rscore[o] = best;
        wk.rcls[o] = bc;
    } else {
        const size_t o = (size_t)b * NMS_CAP + np;
        ((float4 *)wk.dbox)[o] = bx4;
        wk.dscore[o] = best;
        wk.dcls[o] = bc;
    }
    if (p.cand_score) {      // full per-anchor tap (parity tests)
        p.cand_score[(size_t)b * N + n] = best;
        p.cand_cls[(size_t)b * N + n] = bc;
        *(float4 *)(p.cand_box + ((size_t)b * N + n) * 4) = bx4;
    }
}

// ---- compact_kernel (heads with more than CAP anchors): one workgroup per image walks the anchors in the reference's
// anchor-index order n (level, cell, anchor), keeps those with score >= conf_thresh, in order, up to CAP.
__global__ __launch_bounds__(1024) void compact_kernel(const HeadParams p, const HeadWork wk) {
    __shared__ int wsum[16];
    __shared__ int base_s;
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int A = p.A;
    const int HW0 = p.lev[0].Hs * p.lev[0].Ws, N0 = HW0 * A;
    const int HW1 = p.nlev > 1 ? p.lev[1].Hs * p.lev[1].Ws : 0, N1 = N0 + HW1 * A;
    const int HW2 = p.nlev > 2 ? p.lev[2].Hs * p.lev[2].Ws : 0;
    const int N = N1 + HW2 * A;
    if (tid == 0) base_s = 0;
    __syncthreads();
    const float4 *rb = (const float4 *)wk.rbox + (size_t)b * wk.rstride;
    const float *rs = wk.rscore + (size_t)b * wk.rstride;
    const int *rc = wk.rcls + (size_t)b * wk.rstride;
    float4 *db = (float4 *)wk.dbox + (size_t)b * NMS_CAP;
    float *ds = wk.dscore + (size_t)b * NMS_CAP;
    int *dc = wk.dcls + (size_t)b * NMS_CAP;
    int overflow = 0;
    for (int n0 = 0; n0 < N; n0 += 1024) {
        const int n = n0 + tid;
        bool keep = false;
        int np = 0;
        if (n < N) {
            const int lv = (n >= N0 ? 1 : 0) + (n >= N1 ? 1 : 0);
            const int lbase = lv == 0 ? 0 : (lv == 1 ? N0 : N1), HWl = lv == 0 ? HW0 : (lv == 1 ? HW1 : HW2);
            const int cell = (n - lbase) / A, a = (n - lbase) % A;
            np = lbase + a * HWl + cell;                     // where decode_kernel put anchor n
            keep = rs[np] >= p.conf_thresh;
        }
        const unsigned long long m = __ballot(keep);
        const int before = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wsum[wave] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int w = 0; w < wave; ++w) off += wsum[w];
        const int pos = off + before;
        if (keep) {
            if (pos < NMS_CAP) { db[pos] = rb[np]; ds[pos] = rs[np]; dc[pos] = rc[np]; }
            else overflow = 1;
        }
        __syncthreads();
        if (tid == 0) { int t = 0; for (int w = 0; w < 16; ++w) t += wsum[w]; base_s += t; }
        __syncthreads();
    }
    if (__syncthreads_or(overflow) && tid == 0) wk.ovf[b] = 1;
    if (tid == 0) wk.rcount[b] = min(base_s, NMS_CAP);
}


__global__ __launch_bounds__(1024) void head_kernel(const HeadParams p, const HeadWork wk) {
    __shared__ int hist[NMS_CAP];          // bin counts -> bin starts
    __shared__ int wsum[16];
    __shared__ unsigned int sstat[MAXG][4];
    __shared__ int ntiny_s;
    const int b = blockIdx.x, tid = threadIdx.x;
    NSTAMP(0, blockIdx.x, 0);
    const int A = p.A;
    const int HW0 = p.lev[0].Hs * p.lev[0].Ws, N0 = HW0 * A;
    const int HW1 = p.nlev > 1 ? p.lev[1].Hs * p.lev[1].Ws : 0, N1 = N0 + HW1 * A;
    const int HW2 = p.nlev > 2 ? p.lev[2].Hs * p.lev[2].Ws : 0;
    const bool compacted = wk.rbox != nullptr;             // anchors already thresholded and in anchor-index order
    const int N = compacted ? wk.rcount[b] : N1 + HW2 * A;
    const int HWb = p.Hb * p.Wb;
    for (int i = tid; i < NMS_CAP; i += 1024) hist[i] = 0;
    if (tid < MAXG) {
        sstat[tid][0] = 0u;                 // wmax
        sstat[tid][1] = 0u;                 // hmax
        sstat[tid][2] = 0x7f7fffffu;        // amin
        sstat[tid][3] = 0u;                 // amax
    }
    if (tid == 0) ntiny_s = 0;
    if (tid < 2) wk.nedges[b * 2 + tid] = 0;
    __syncthreads();
    NSTAMP(0, blockIdx.x, 1);

    float box[4][4], score[4];
    int cls[4], orig[4], key[4], rk[4], ktu[4], kb[4] = {0, 0, 0, 0};
    bool valid[4];
    int tkt = -1;                                          // group of this thread's candidates (mixed: more than one)
    bool mixed = false;
    unsigned int tst[4] = {0u, 0u, 0x7f7fffffu, 0u};
    {
        // this thread's four anchors np = 4 tid .. 4 tid + 3 of the decode (NMS_CAP is a multiple of 4)
        const size_t o4 = (size_t)b * NMS_CAP + (size_t)tid * 4;
        float4 d[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) d[u] = ((const float4 *)wk.dbox)[o4 + u];
        const float4 sc4 = *(const float4 *)(wk.dscore + o4);
        const int4 cl4 = *(const int4 *)(wk.dcls + o4);
        const float scs[4] = {sc4.x, sc4.y, sc4.z, sc4.w};
        const int cls4[4] = {cl4.x, cl4.y, cl4.z, cl4.w};
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            box[u][0] = d[u].x; box[u][1] = d[u].y; box[u][2] = d[u].z; box[u][3] = d[u].w;
            score[u] = scs[u];
            cls[u] = cls4[u];
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int np = tid * 4 + u;          // (level, anchor, cell) enumeration
        valid[u] = false;
        orig[u] = 0;
        key[u] = 0;
        rk[u] = 0;
        ktu[u] = 0;
        if (np < N) {
            const int lv = (np >= N0 ? 1 : 0) + (np >= N1 ? 1 : 0);
            const int lbase = lv == 0 ? 0 : (lv == 1 ? N0 : N1);
            const int npl = np - lbase, HWl = lv == 0 ? HW0 : (lv == 1 ? HW1 : HW2);
            const int a = npl / HWl, cell = npl % HWl;
            orig[u] = compacted ? np : lbase + cell * A + a;     // compacted: position = rank in anchor-index order
            valid[u] = score[u] >= p.conf_thresh;
            if (valid[u]) {
                // candidate group: the anchor type, or (group_by_area) the octave of the clamped box's
                // area -- boxes with IoU > thr have areas within 1/thr of each other, so a group only
                // meets its neighbours; with wide-ranging exp(tw) the anchor says little about the
                // size.  Per-group extents, bin of the clamped centre inside the group.
                const float w = box[u][2] - box[u][0], h = box[u][3] - box[u][1], ar = w * h;
                const int ex = (int)((__float_as_uint(ar) >> 23) & 0xffu) - 127;     // floor(log2 area), area <= 1
                // p.cls_groups (heads with 3 .. 32 classes, round 4): the group IS the class -- per-class NMS never pairs two
                // classes, so the pair walk of a candidate only visits its own class's bins (1 / C of the partners); the extents are
                // then kept over ALL candidates (slot 0: one butterfly per wave): the window is bounded by the candidate's own
                // size through w_j <= w_i / thr anyway
                const int kt = p.cls_groups ? min(max(cls[u], 0), p.C - 1) : (p.group_by_area ? min(NGROUP - 1, max(0, -ex - 1)) : lv * A + a);
                const int ks = p.cls_groups ? 0 : kt;                 // where its extents are accumulated
                const float ccx = 0.5f * (box[u][0] + box[u][2]), ccy = 0.5f * (box[u][1] + box[u][3]);
                const int bx = min(p.Wb - 1, max(0, (int)(ccx * (float)p.Wb)));
                const int by = min(p.Hb - 1, max(0, (int)(ccy * (float)p.Hb)));
                key[u] = kt * HWb + by * p.Wb + bx;
                ktu[u] = kt;
                rk[u] = atomicAdd(&hist[key[u]], 1);
                // per-group extents: accumulated per thread, merged per wave below (four same-address LDS atomics per
                // candidate serialised the whole workgroup: 29 k cycles, bank-conflict share 0.89)
                if (tkt >= 0 && tkt != ks) mixed = true;
                tkt = ks;
                kb[u] = ks;
                tst[0] = max(tst[0], __float_as_uint(w));
                tst[1] = max(tst[1], __float_as_uint(h));
                tst[2] = min(tst[2], __float_as_uint(ar));
                tst[3] = max(tst[3], __float_as_uint(ar));
            }
        }
    }
    {
        // a wave covers 256 consecutive positions of the (level, anchor, cell) enumeration: one group, except at the few
        // group boundaries.  Uniform wave: butterfly reduction, four atomics per wave; otherwise per-candidate atomics.
        const unsigned long long vm = __ballot(tkt >= 0);
        if (vm) {
            const int first = __ffsll((long long)vm) - 1;
            const int kt0 = __shfl(tkt, first, 64);
            const bool uni = __all(tkt < 0 || (tkt == kt0 && !mixed));
            if (uni) {
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    tst[0] = max(tst[0], (unsigned int)__shfl_xor((int)tst[0], o, 64));
                    tst[1] = max(tst[1], (unsigned int)__shfl_xor((int)tst[1], o, 64));
                    tst[2] = min(tst[2], (unsigned int)__shfl_xor((int)tst[2], o, 64));
                    tst[3] = max(tst[3], (unsigned int)__shfl_xor((int)tst[3], o, 64));
                }
                if ((tid & 63) == first) {
                    atomicMax(&sstat[kt0][0], tst[0]);
                    atomicMax(&sstat[kt0][1], tst[1]);
                    atomicMin(&sstat[kt0][2], tst[2]);
                    atomicMax(&sstat[kt0][3], tst[3]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (valid[u]) {
                        const float w = box[u][2] - box[u][0], h = box[u][3] - box[u][1], ar = w * h;
                        atomicMax(&sstat[kb[u]][0], __float_as_uint(w));
                        atomicMax(&sstat[kb[u]][1], __float_as_uint(h));
                        atomicMin(&sstat[kb[u]][2], __float_as_uint(ar));
                        atomicMax(&sstat[kb[u]][3], __float_as_uint(ar));
                    }
                }
            }
        }
    }
    __syncthreads();
    NSTAMP(0, blockIdx.x, 2);
    // ---- exclusive scan of the bin counts: 4 consecutive bins per thread
    const int lane = tid & 63, wave = tid >> 6;
    int c4[4], mine = 0;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        c4[u] = hist[tid * 4 + u];
        mine += c4[u];
    }
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int t = __shfl_up(incl, o, 64);
        if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    NSTAMP(0, blockIdx.x, 3);
    if (tid == 0) {
        int s = 0;
        for (int w = 0; w < 16; ++w) { const int t = wsum[w]; wsum[w] = s; s += t; }
        wk.count[b] = s;
    }
    __syncthreads();
    NSTAMP(0, blockIdx.x, 4);
    int run = wsum[wave] + incl - mine;
    int *bs = wk.binstart + (size_t)b * (NMS_CAP + 8);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        hist[tid * 4 + u] = run;
        bs[tid * 4 + u] = run;
        run += c4[u];
    }
    if (tid == 1023) bs[NMS_CAP] = run;
    __syncthreads();
    NSTAMP(0, blockIdx.x, 5);
    // ---- scatter into (anchor, bin) order
    float *cb = wk.cbox + (size_t)b * NMS_CAP * 4;
    float *cs = wk.cscore + (size_t)b * NMS_CAP;
    int *cc = wk.ccls + (size_t)b * NMS_CAP;
    int *co = wk.corig + (size_t)b * NMS_CAP;
    int *tl = wk.tiny + (size_t)b * NMS_CAP;
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (valid[u]) {
            const int pos = hist[key[u]] + rk[u];
            *(float4 *)(cb + (size_t)pos * 4) = make_float4(box[u][0], box[u][1], box[u][2], box[u][3]);
            cs[pos] = score[u];
            cc[pos] = cls[u];
            co[pos] = orig[u];
            wk.ctype[(size_t)b * NMS_CAP + pos] = ktu[u];
            if ((box[u][2] - box[u][0]) * (box[u][3] - box[u][1]) < AREA_MIN) tl[atomicAdd(&ntiny_s, 1)] = pos;
        }
    }
    if (tid < MAXG) {                                       // class groups all take slot 0's (global) extents
        float *as = wk.astat + ((size_t)b * MAXG + tid) * 4;
        const int src = p.cls_groups ? (tid < p.C ? 0 : MAXG - 1) : tid;       // (slot MAXG - 1 is never written: "empty")
#pragma unroll
        for (int k = 0; k < 4; ++k) as[k] = __uint_as_float(sstat[src][k]);
    }
    __syncthreads();
    NSTAMP(0, blockIdx.x, 6);
    if (tid == 0) wk.ntiny[b] = ntiny_s;
    NSTAMP(0, blockIdx.x, 7);
}

// ---- the reference's suppression test (slim_yolo_v2.py:159-171), same class assumed
__device__ __forceinline__ bool suppresses_exact(const float4 a, float area_a, const float4 c, float area_c, float thr) {
    const float xx1 = fmaxf(a.x, c.x), yy1 = fmaxf(a.y, c.y);
    const float xx2 = fminf(a.z, c.z), yy2 = fminf(a.w, c.w);
    const float w = fmaxf(1e-28f, xx2 - xx1), h = fmaxf(1e-28f, yy2 - yy1);
    const float inter = w * h;
    const float ovr = inter / (area_a + area_c - inter);
    return !(ovr <= thr);
}
// same predicate; the correctly rounded division is only issued when the reciprocal estimate
// lands within 8 ulp-ish of the threshold (or the union is degenerate)
__device__ __forceinline__ bool suppresses(const float4 a, float area_a, const float4 c, float area_c, float thr,
                                           float thr_hi, float thr_lo) {
    const float xx1 = fmaxf(a.x, c.x), yy1 = fmaxf(a.y, c.y);
    const float xx2 = fminf(a.z, c.z), yy2 = fminf(a.w, c.w);
    const float w = fmaxf(1e-28f, xx2 - xx1), h = fmaxf(1e-28f, yy2 - yy1);
    const float inter = w * h;
    const float den = area_a + area_c - inter;
    const float q = inter * __builtin_amdgcn_rcpf(den);
    const bool normal = den > 1e-30f && den < 1e30f;
    const bool hi = normal && q > thr_hi, lo = normal && q < thr_lo;
    bool r = hi;
    if (__any(!hi && !lo)) {
        const bool ex = !(inter / den <= thr);
        r = (hi || lo) ? hi : ex;
    }
    return r;
}

// Exact pruning bounds.  For boxes with IoU > thr:  overlap_x > thr*max(w) hence
// |dcx| < (wi+wj)/2 - thr*max(wi,wj) <= (1-thr)*(wi+wj)/2 (same in y), and min(area)/max(area) > thr.
// With a 0.1% margin the fp32 evaluation of the reference formula cannot land above thr either,
// provided the union is not degenerate (area sum >= 1e-10) and thr >= 1e-4.
// grid (CAP/1024, batch), 1024 threads: one thread per candidate i.  Every unordered pair is
// visited once, from its lower compact position: partners q > i inside the window.  The image's
// candidates, their groups and the bin starts are staged in LDS so the walk never waits on global memory.
//
// The walk is written for few instructions and few branches (the first version spent its time in control flow: 290
// branches / 265 exec-mask saves in 4 650 instructions, 3-4 k cycles per trip of four partners):
//   * a trip loads four partners and evaluates the whole predicate branch-free; only a quotient within 8e-6 of the
//     threshold (rare) takes a wave-uniform branch to the IEEE division;
//   * hits go to a per-WAVE edge buffer in LDS: position = wave-uniform count + rank in the ballot, no atomics; a full
//     buffer (and every wave at its end) is flushed to the image's global list with one global atomic;
//   * the next bin row's range is read while the current row is walked.
#define WAVE_EDGE_CAP (WG_EDGE_CAP / 16)
#define LDS_EDGE_CAP Y355_HEAD_EDGE_CAP   // most edges of one image resolve_emit_kernel's in-edge lists hold (the sorted walk takes over beyond)
#define PAIRS_ABORT_EDGES LDS_EDGE_CAP
__device__ __forceinline__ float raw_max(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float raw_min(float a, float b) {
    float r;
    asm("v_min_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
#ifndef Y355_PAIRS_G
#define Y355_PAIRS_G 2             // workgroups per image: 4 finish a batch no sooner (the heavy waves set the time) but hold twice the CUs,
                                   // which the convolutions of the other streams cannot use meanwhile (3-stream: 227 k -> 239 k img/s);
                                   // round 4: ONE while several handles share the GPU (HeadParams::pairs_wgs, set by the engine in its
                                   // throughput mode: 283.1 k -> 286.9 k img/s; a handle alone keeps two: 195 k vs 189 k one stream)
#endif
template <bool FAST>
__global__ __launch_bounds__(1024) void pairs_kernel(const HeadParams p, const HeadWork wk, float thr) {
    extern __shared__ __attribute__((aligned(16))) char plds[];
    float4 *sbox = (float4 *)plds;                                   // [CAP]
    int2 *scls = (int2 *)(plds + NMS_CAP * 16);                      // [CAP] (class | group << 16, area of the box as computed here)
    int *sbin = (int *)(plds + NMS_CAP * 24);                        // [CAP + 8]
    unsigned int *sedge = (unsigned int *)(plds + NMS_CAP * 24 + (NMS_CAP + 8) * 4);   // [16][WAVE_EDGE_CAP]
    __shared__ float as[MAXG * 4];                         // per-group extents
    const int b = blockIdx.y;
    NSTAMP(1, (blockIdx.y * gridDim.x + blockIdx.x), 0);
    const int M = wk.count[b];
    // L lanes share a candidate (each takes every L-th group of four partners of a bin row) when the image has few
    // candidates: 845 of them are 14 waves, 3-4 per workgroup, one per SIMD -- with L = 4 every SIMD gets four
    const int G = (int)gridDim.x;
    // round 4: the walk is VALU-bound and a wave's time is its trips (stamps: 2-4 k cycles per trip of four partners with the
    // workgroup's 16 waves active); the 64-candidate runs differ by 10x in trips (YOLOv3tiny: median 8, maximum 142), so an image
    // with few candidates is cut into more, lighter runs: L lanes per candidate as long as M * L stays within PAIRS_LANES_BUDGET
#ifndef PAIRS_LANES_BUDGET
#define PAIRS_LANES_BUDGET 8192            // measured: 2048 / 4096 / 8192 / 16384 -> YOLOv3tiny NMS 146 / 91 / 86 / 94 us, SlimYOLOv2 fp32 147 / 148 / 114 / 127, q_bf headline 74.5 / 74.2 / 74.6 / 97.3
#endif
    int L = 1;
    while (L < 8 && M * 2 * L <= PAIRS_LANES_BUDGET * G / 2) L *= 2;
    const int nruns = (M * L + 63) >> 6;                    // 64-lane runs of the image
    if ((int)blockIdx.x >= nruns) return;
    const int tid = threadIdx.x, lane = tid & 63;
    const int A = p.cls_groups ? p.C : (p.group_by_area ? NGROUP : p.A * p.nlev), Ws = p.Wb, Hs = p.Hb, HW = Hs * Ws;   // candidate groups, bin grid
    {
        const float4 *cbx4 = (const float4 *)(wk.cbox + (size_t)b * NMS_CAP * 4);
        const int *ccl = wk.ccls + (size_t)b * NMS_CAP;
        const int *cty = wk.ctype + (size_t)b * NMS_CAP;
        const int *bs = wk.binstart + (size_t)b * (NMS_CAP + 8);
        // all loads of a thread issued together (one global latency, not one per iteration)
        float4 vb[NMS_CAP / 1024];
        int vc[NMS_CAP / 1024], vt[NMS_CAP / 1024], vs[NMS_CAP / 1024 + 1];
#pragma unroll
        for (int u = 0; u < NMS_CAP / 1024; ++u) {
            const int q = tid + u * 1024;
            vb[u] = make_float4(0.f, 0.f, 0.f, 0.f);
            vc[u] = 0; vt[u] = 0;
            if (q < M) { vb[u] = cbx4[q]; vc[u] = ccl[q]; vt[u] = cty[q]; }
        }
#pragma unroll
        for (int u = 0; u <= NMS_CAP / 1024; ++u) {
            const int q = tid + u * 1024;
            vs[u] = (q <= A * HW) ? bs[q] : 0;
        }
#pragma unroll
        for (int u = 0; u < NMS_CAP / 1024; ++u) {
            const int q = tid + u * 1024;
            if (q < M) {
                sbox[q] = vb[u];
                scls[q] = make_int2((vc[u] & 0xffff) | (vt[u] << 16), __float_as_int((vb[u].z - vb[u].x) * (vb[u].w - vb[u].y)));
            }
        }
#pragma unroll
        for (int u = 0; u <= NMS_CAP / 1024; ++u) {
            const int q = tid + u * 1024;
            if (q <= A * HW) sbin[q] = vs[u];
        }
        if (tid < MAXG * 4) as[tid] = wk.astat[(size_t)b * MAXG * 4 + tid];
    }
    __syncthreads();
    NSTAMP(1, (blockIdx.y * gridDim.x + blockIdx.x), 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr bool fast = FAST;                            // 1e-4 <= thr < 1e4: the pruning bounds hold (host-checked)
    const float kr = (1.0f - thr) * 0.5f * PRUNE_MARGIN, thr_lo = thr * 0.999f, inv_thr_lo = 1.0f / thr_lo;
    const float q_hi = thr * (1.0f + 8e-6f), q_lo = thr * (1.0f - 8e-6f);
    bool lost = false;
    unsigned int *wedge = sedge + wave * WAVE_EDGE_CAP;
    unsigned int *ge = wk.edges + (size_t)b * EDGE_CAP;
    int wcount = 0;                                        // wave-uniform: edges in this wave's buffer
    int dbg_trips = 0;
    (void)dbg_trips;
    // Early abort (round 4): resolve_emit_kernel settles an image by rounds over its edge list only when the list has at most
    // LDS_EDGE_CAP entries; beyond that (or on the overflow flag) it takes the sorted walk, which evaluates the predicate itself
    // and never reads the list.  An image whose candidates nearly all overlap (the random-weight fp32 fixtures: 845 candidates,
    // ~300 k suppressing pairs) used to have every one of those pairs tested, buffered and written here first -- 214 us of
    // pairs_kernel per batch for a list nobody read.  The running total comes back from every flush's atomic: the wave that sees
    // it pass the limit raises the overflow flag and stops, the others stop at their next flush or run.
    bool aborted = false;                                  // wave-uniform
    auto flush = [&]() {                                   // wave-uniform call
        int g = 0;
        if (lane == 0) g = atomicAdd(&wk.nedges[b * 2], wcount);
        g = __builtin_amdgcn_readfirstlane(g);
        if (g + wcount > PAIRS_ABORT_EDGES) {
            aborted = true;
            lost = true;
            if (lane == 0) __hip_atomic_store(&wk.nedges[b * 2 + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            for (int e = lane; e < wcount; e += 64) ge[g + e] = wedge[e];
        }
        wcount = 0;
    };
    // 64-lane runs are dealt round-robin to the image's workgroups and their waves: neighbours in (group, bin) order have
    // windows of similar size, so a wave stays uniform while every workgroup gets the same mix of cheap and expensive runs
#if defined(Y355_ABL_NMS) && (Y355_ABL_NMS & 2)
    if (false)                                      // timing ablation (WRONG RESULTS): no pair walk
#endif
    for (int run = wave * G + (int)blockIdx.x; run < nruns; run += 16 * G) {
    {   // the overflow flag is read by lane 0 and broadcast: `aborted` steers scalar state (wcount, the buffer index), so it must be
        // wave-uniform by construction, not by the luck of 64 lanes loading the same word at the same moment
        int fl = 0;
        if (lane == 0) fl = __hip_atomic_load(&wk.nedges[b * 2 + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (aborted || __builtin_amdgcn_readfirstlane(fl) != 0) { aborted = true; break; }
    }
    const int slot = (run << 6) + lane;
    const int i = slot / L, sub = slot % L;
    const bool vi = i < M;
    const float4 bi = vi ? sbox[i] : make_float4(0, 0, 0, 0);
    const int cti = vi ? scls[i].x : -1;                   // class | group << 16: equal classes <=> equal low halves
    const int ci = cti & 0xffff;
    const float wi = bi.z - bi.x, hi = bi.w - bi.y, ai = wi * hi;
    const float cxi = 0.5f * (bi.x + bi.z), cyi = 0.5f * (bi.y + bi.w);
    // the reference's predicate (slim_yolo_v2.py:159-171) for partner q of candidate i, branch-free; `ok` = q is a real
    // partner of this lane's trip
    auto test = [&](int q, bool ok, const float4 bj, const int2 ctj) {
        // No pruning test here: the window already holds only bins a suppressor can sit in, and a pair the bounds would rule
        // out evaluates to ovr <= thr in the formula below as well (that is what the bounds prove) -- in branch-free code the
        // extra test was 25 instructions that saved none.  The partner's area comes from LDS (computed once per candidate).
        const float aj = __int_as_float(ctj.y);
        const bool cand = ok & (q > i) & ((ctj.x & 0xffff) == ci);      // (&, not &&: no branch, no bool -> int -> bool round trip)
        // v_max / v_min issued as they are: hipcc quiets BOTH operands of every fmaxf / fminf first (8 extra v_max per test);
        // on numbers -- the coordinates are clamped to [0, 1] by the decode -- the bare instructions give the same result
        const float xx1 = raw_max(bi.x, bj.x), yy1 = raw_max(bi.y, bj.y);
        const float xx2 = raw_min(bi.z, bj.z), yy2 = raw_min(bi.w, bj.w);
        const float iw = raw_max(1e-28f, xx2 - xx1), ih = raw_max(1e-28f, yy2 - yy1);
        const float inter = iw * ih, den = ai + aj - inter;
        const float qv = inter * __builtin_amdgcn_rcpf(den);
        const bool sure = fast & (den > 1e-30f) & ((qv > q_hi) | (qv < q_lo));     // (den <= 2: the decode clamps every box to [0, 1]; a NaN fails every comparison and takes the exact path)
        bool hit = cand & sure & (qv > q_hi);
        if (__any(cand & !sure)) {                         // rare: the correctly rounded quotient decides
            if (cand & !sure) hit = !(inter / den <= thr);
        }
        const unsigned long long m = __ballot(hit);
        if (m) {
            if (hit) {
                const int e = wcount + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
                wedge[e] = ((unsigned int)i << 12) | (unsigned int)q;
            }
            wcount += (int)__popcll(m);
        }
    };
    // Every loop below is WAVE-UNIFORM (bounds through __any, lanes without work carry ok = false): the edge count of the
    // wave's buffer is a scalar only if all lanes see every update.
    auto walk = [&](int q0, int q1) {                      // partners [q0, q1) of this candidate (empty: q0 >= q1)
        for (int q = q0 + 4 * sub; !aborted && __any(q < q1); q += 4 * L) {
#ifdef Y355_EXPERIMENTS
            ++dbg_trips;
#endif
            float4 bj[4];
            int2 cj[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int qq = max(min(q + u, q1 - 1), 0);
                bj[u] = sbox[qq];
                cj[u] = scls[qq];
            }
            if (wcount > WAVE_EDGE_CAP - 256) flush();     // room for this trip's 4 x 64 hits
#pragma unroll
            for (int u = 0; u < 4; ++u) test(q + u, q + u < q1, bj[u], cj[u]);
        }
    };
    {
        // own group and bin row (same formula as head_kernel); lanes past M walk nothing (a_i = A)
        const int a_i = vi ? (cti >> 16) : A;
        const int by_i = min(Hs - 1, max(0, (int)(cyi * (float)Hs)));
        // candidates are sorted by group: lane 0 holds the run's lowest one, and lower groups only hold positions < i
        const int a_lo = __builtin_amdgcn_readfirstlane(a_i);
        for (int a2 = a_lo; a2 < A && !aborted; ++a2) {
            const float amin = as[a2 * 4 + 2], amax = as[a2 * 4 + 3];
            if (amax < amin) continue;                         // no candidate of this group (uniform)
            const bool prune = fast && (ai + amin >= AREA_MIN);
            bool act = p.cls_groups ? a2 == a_i : a2 >= a_i;       // class groups: only my own class
            if (prune && (ai <= thr_lo * amin || amax <= thr_lo * ai)) act = false;    // area ratio rules the group out
            if (!__any(act)) continue;
            int bx0 = 0, bx1 = Ws - 1, by0 = 0, by1 = Hs - 1;
            if (prune) {
                // a partner j with IoU > thr has w_j <= w_i / thr and h_j <= h_i / thr (the intersection is at most w_i h_j, the
                // union at least w_j h_j): the group's widest / tallest box bounds the window only up to that (round 4: an area
                // octave of the fp32 heads holds 1.0 x 0.06 boxes beside 0.25 x 0.25 ones, and every narrow candidate walked the
                // whole width of the bin grid)
                const float wmax = fminf(as[a2 * 4 + 0], wi * inv_thr_lo), hmax = fminf(as[a2 * 4 + 1], hi * inv_thr_lo);
                const float rx = kr * (wi + wmax) + PRUNE_EPS, ry = kr * (hi + hmax) + PRUNE_EPS;
                bx0 = max(0, (int)floorf((cxi - rx) * (float)Ws));
                bx1 = min(Ws - 1, (int)floorf((cxi + rx) * (float)Ws));
                by0 = max(0, (int)floorf((cyi - ry) * (float)Hs));
                by1 = min(Hs - 1, (int)floorf((cyi + ry) * (float)Hs));
            }
            if (a2 == a_i) by0 = max(by0, by_i);               // earlier bin rows of my group are < i
            if (!act || by0 > by1 || bx0 > bx1) { by0 = 0; by1 = -1; bx0 = 0; bx1 = 0; }   // an empty window, valid addresses
            if (!__any(by0 <= by1)) continue;
            int k0 = a2 * HW + by0 * Ws;
            int ra = sbin[k0 + bx0], rb = sbin[k0 + bx1 + 1];
            for (int by = by0; __any(by <= by1); ++by) {
                // the next row's range is in flight while this row is walked (past the window it is read and dropped)
                const int kn = min(k0 + Ws, A * HW - Ws);
                const int nra = sbin[kn + bx0], nrb = sbin[kn + bx1 + 1];
                const bool row = by <= by1;
                walk(row ? max(ra, i + 1) : 0, row ? rb : 0);
                ra = nra; rb = nrb;
                k0 = kn;
            }
        }
        {                                                          // degenerate boxes see each other
            const bool tiny_i = fast && vi && ai < AREA_MIN;
            if (!aborted && __any(tiny_i)) {                       // (an aborted image's list is never read: no more tests, no more flushes)
                const int nt = wk.ntiny[b];
                const int *tl = wk.tiny + (size_t)b * NMS_CAP;
                for (int t0 = 0; t0 < nt && !aborted; ++t0) {
                    if (wcount > WAVE_EDGE_CAP - 64) flush();
                    const int q = tl[t0];
                    test(q, tiny_i && sub == 0, sbox[q], scls[q]);
                }
            }
        }
    }
    }   // runs
#ifdef Y355_EXPERIMENTS
    if (wk.stamps) {                                   // per-wave end of the walk and the wave-wide trip count (max over lanes)
        int mt = dbg_trips;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mt = max(mt, __shfl_xor(mt, o, 64));
        const int wgl = blockIdx.y * gridDim.x + blockIdx.x;
        if (lane == 0 && wgl < 128)
            wk.stamps[(3 * 256 + wgl * 2 + (tid >> 9)) * 8 + ((tid >> 6) & 7)] =
                (__builtin_amdgcn_s_memtime() & 0xffffffffffull) | ((unsigned long long)mt << 40);
    }
#endif
    if (wcount > 0 && !aborted) flush();
    if (lost) __hip_atomic_store(&wk.nedges[b * 2 + 1], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    NSTAMP(1, (blockIdx.y * gridDim.x + blockIdx.x), 7);
}
#define PAIRS_LDS (NMS_CAP * 24 + (NMS_CAP + 8) * 4 + WG_EDGE_CAP * 4)

// ---- resolve_emit_kernel: one workgroup per image.
// Greedy NMS = for every candidate, "kept unless an EARLIER (score desc, anchor index asc) kept
// candidate suppresses it".  With the suppressing pairs known, that is settled without sorting, by
// rounds over the edge list held in LDS (each thread owns a strided set of slots and compacts it in
// place):
//   A  for every live edge (a earlier -> b): a kept -> b is dead, edge retires; a dead -> edge
//      retires; a undecided -> b is blocked this round;
//   B  every undecided candidate that is neither dead nor blocked is kept.
// A candidate is decided one round after its last earlier neighbour, so the number of rounds is the
// longest chain of alternating decisions (16-17 on the benchmark's images), each a few hundred
// cycles.  Classes need no special handling: edges only join candidates of one class.
// Survivors are then written in anchor-index order into the padded outputs.
// Fallback (an edge list overflowed, or more edges than the LDS list holds): the textbook walk over
// the candidates sorted by score, with the reference's predicate evaluated on the fly.
// Round 5: a node-centric ASYNCHRONOUS settle (in-edge lists per candidate, every thread polling its own candidates'
// earlier neighbours, no block barriers) was built bit-exact and measured 53.7 us (all four candidates of a thread per trip) and
// 112.8 us (one candidate per trip, neighbour indices in registers) against these rounds' 40.6: a decision still travels one
// WAVE TRIP per link of a suppression chain, and a trip of sixteen polling waves costs what a round costs
// (scratch/nms_experiments/head_nms_r5_async_settle.hip, profiles/r05_notes.md section 2).
#define RE_REG 7                    // edge slots per thread held in registers (7168 edges)
#define RE_NONE 0x00ffffffu          // a retired / absent edge: (4095, 4095), never a real pair (i < q)
#ifndef RE_DENSE_AFTER
#define RE_DENSE_AFTER 3            // rounds over the register slots before the waves pack their live edges (0: never)
#endif
#ifndef RE_TAIL
#define RE_TAIL 512                 // live edges at which ONE wave finishes the rounds alone, without barriers (a multiple of 64)
#endif

__global__ __launch_bounds__(1024) void resolve_emit_kernel(const HeadParams p, const HeadWork wk, float thr) {
    __shared__ __attribute__((aligned(16))) unsigned int sedge[LDS_EDGE_CAP - RE_REG * 1024];   // edges past the register slots; sort keys of the fallback walk
    __shared__ __attribute__((aligned(16))) unsigned char state[NMS_CAP];        // 0 undecided, 1 kept, 2 dead
    __shared__ __attribute__((aligned(16))) unsigned char blocked[3][NMS_CAP];   // round r writes [r % 3], reads [(r - 1) % 3], clears [(r + 1) % 3]
    __shared__ unsigned int skey[2 * NMS_CAP];        // NMS order key of a candidate: (score bits, ~anchor index); later the emit scratch
    __shared__ unsigned long long keepn[64];          // survivors by anchor index
    __shared__ int pend[3];
    __shared__ int wcnt[16];
    __shared__ int wbase[64];
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63;
    NSTAMP(2, blockIdx.x, 0);
    const int M = wk.count[b];
    const float *cb = wk.cbox + (size_t)b * NMS_CAP * 4;
    const float *cs = wk.cscore + (size_t)b * NMS_CAP;
    const int *cc = wk.ccls + (size_t)b * NMS_CAP;
    const int *co = wk.corig + (size_t)b * NMS_CAP;
    const int ne = wk.nedges[b * 2];
    const bool brute = wk.nedges[b * 2 + 1] != 0 || ne > LDS_EDGE_CAP;
    const unsigned int *ge = wk.edges + (size_t)b * EDGE_CAP;
    for (int pos = tid; pos < NMS_CAP; pos += 1024) {
        state[pos] = (pos < M && brute) ? 0 : 1;          // kept unless it is an endpoint of an edge (below)
        blocked[0][pos] = 0;
        blocked[1][pos] = 0;
        blocked[2][pos] = 0;
        // scores are non-negative floats: their bit patterns order like the values.  Staged once, coalesced: the
        // orientation loop below used to gather cs / co from global memory per edge (15 k cycles of dependent loads)
        skey[2 * pos] = pos < M ? __float_as_uint(cs[pos]) : 0u;
        skey[2 * pos + 1] = pos < M ? (unsigned int)co[pos] : 0u;
    }
    if (tid < 64) keepn[tid] = 0ull;
    if (tid < 3) pend[tid] = 0;
    __syncthreads();
    if (!brute) {
        // The first RE_REG entries of a thread's edge slots (slot k of thread tid = list entry tid + 1024 k) live in registers,
        // the rest (only images with more than 7168 suppressing pairs have any) in LDS; the candidate states in LDS.
        // Round 6: ONE barrier and one pass per round (rounds 2-5: two of each -- A: edges mark, B: every thread settles its own
        // four candidates).  B is never materialised: "undecided and not blocked in the previous round" IS kept -- every live
        // in-edge of such a candidate retired in that round (an in-edge whose earlier endpoint was still undecided would have
        // blocked it, one whose earlier endpoint was kept would have killed it) -- so the edge walk of round r reads, for its
        // earlier endpoint a, state[a] and blocked[(r - 1) % 3][a] and acts on
        //     a dead                                    -> the edge is void
        //     a kept, or undecided and unblocked before -> the later endpoint dies, the edge retires
        //     a undecided and blocked before            -> the later endpoint is blocked in round r
        // Three blocked buffers rotate (written / read / cleared by the owners of its words); orientation is round 0: every edge
        // blocks its later endpoint.  At the end a candidate is kept unless it is dead.  Same decisions as the two-pass rounds,
        // one round earlier each; measured 39.0 -> 2x.x us on the headline fixture (profiles/r06_notes.md).
        const int kmax = __builtin_amdgcn_readfirstlane((ne + 1023) >> 10);
        unsigned int ed[RE_REG];
        auto orient = [&](unsigned int pq) {
            // orient the pair by the NMS order (score desc, anchor index asc); both endpoints become undecided, the later one blocked
            const int i = (int)(pq >> 12), q = (int)(pq & 0xfffu);
            const uint2 ki = *(const uint2 *)&skey[2 * i], kq = *(const uint2 *)&skey[2 * q];
            const bool i_first = ki.x > kq.x || (ki.x == kq.x && ki.y < kq.y);
            state[i] = 0;
            state[q] = 0;
            blocked[0][i_first ? q : i] = 1;
            return i_first ? pq : (((unsigned int)q << 12) | (unsigned int)i);
        };
#pragma unroll
        for (int k = 0; k < RE_REG; ++k) {
            const int e = tid + k * 1024;
            ed[k] = e < ne ? ge[e] : RE_NONE;
        }
#pragma unroll
        for (int k = 0; k < RE_REG; ++k)
            if (ed[k] != RE_NONE) ed[k] = orient(ed[k]);
        for (int k = RE_REG; k < kmax; ++k) {
            const int e = tid + k * 1024;
            sedge[e - RE_REG * 1024] = e < ne ? orient(ge[e]) : RE_NONE;
        }
        __syncthreads();
        NSTAMP(2, blockIdx.x, 1);
        int nround = 0, rb = 0;                      // rb = nround % 3: the buffer this round's blocks go to
        (void)nround;
        for (;;) {
            ++nround;
            const int pb = rb;                       // the previous round's buffer
            rb = rb == 2 ? 0 : rb + 1;
            const int cb = rb == 2 ? 0 : rb + 1;     // the next round's buffer: cleared now
            unsigned char *bw = blocked[rb];
            const unsigned char *bp = blocked[pb];
            // one step for an edge whose earlier endpoint was read as (state sa, blocked before ba): the edge, or RE_NONE once it
            // retires (the later endpoint's state is not read: an edge whose later endpoint already died only blocks it a few
            // more times, which nobody looks at)
            auto settle = [&](unsigned int e, int sa, int ba) {
                if (e == RE_NONE || sa == 2) return RE_NONE;                       // the earlier endpoint died: the edge is void
                const int c = (int)(e & 0xfffu);
                if (sa == 1 || ba == 0) { state[c] = 2; return RE_NONE; }          // earlier endpoint kept: the later one dies
                bw[c] = 1;                                                          // earlier endpoint still undecided
                return e;
            };
            int live = 0;
            {
                int sa[RE_REG], ba[RE_REG];
#pragma unroll
                for (int k = 0; k < RE_REG; ++k) {                                 // RE_NONE reads candidate 4095: harmless
                    sa[k] = state[ed[k] >> 12];
                    ba[k] = bp[ed[k] >> 12];
                }
#pragma unroll
                for (int k = 0; k < RE_REG; ++k) {
                    ed[k] = settle(ed[k], sa[k], ba[k]);
                    live |= ed[k] != RE_NONE;
                }
            }
            for (int k = RE_REG; k < kmax; ++k) {                                   // rare: the LDS-resident tail
                const unsigned int e = sedge[tid + (k - RE_REG) * 1024];
                if (e != RE_NONE) {
                    const unsigned int e2 = settle(e, state[e >> 12], bp[e >> 12]);
                    sedge[tid + (k - RE_REG) * 1024] = e2;
                    live |= e2 != RE_NONE;
                }
            }
            *(unsigned int *)&blocked[cb][4 * tid] = 0u;
            // "any edge still live?" through one LDS word per round and ONE barrier
            if (live) pend[rb] = 1;
            if (tid == 0) pend[cb] = 0;
            __syncthreads();
            const int again = pend[rb];
#ifdef Y355_EXPERIMENTS
            if (nround == 1) NSTAMP(2, blockIdx.x, 3);
            if (nround == 2) NSTAMP(2, blockIdx.x, 5);
            if (!again && wk.stamps && tid == 0 && blockIdx.x < 256) wk.stamps[((2 * 256 + blockIdx.x) * 8) + 6] = wk.stamps[((2 * 256 + blockIdx.x) * 8) + 5] + nround;
#endif
#if defined(Y355_ABL_NMS) && (Y355_ABL_NMS & 1)
            break;                                  // timing ablation (WRONG RESULTS): one round only
#endif
            if (!again) break;
#if RE_DENSE_AFTER
            // A round over the register slots costs the same whether 6 700 edges are live or 60 (16 waves x 7 slots x ~18
            // instructions: ~2.6 k cycles, measured with either one or two barriers per round -- the SIMDs' issue slots, not the
            // barriers), and most edges retire in the first rounds (headline fixture: 6 772 -> 3 761 -> 2 274 -> 1 236 -> 552 live
            // after rounds 1 / 3 / 5 / 7 / 9 of 16).  After RE_DENSE_AFTER rounds every wave packs its live edges into a private
            // list in LDS and the remaining rounds walk those lists, re-packing in place: a wave's work per round is then
            // proportional to ITS live edges, and a wave whose list ran empty only meets the barrier.
            if (nround == RE_DENSE_AFTER && kmax <= RE_REG) {
                const int wave = tid >> 6;
                unsigned int *mine = sedge + wave * (RE_REG * 64);
                int cnt = 0;
#pragma unroll
                for (int k = 0; k < RE_REG; ++k) {
                    const bool keep = ed[k] != RE_NONE;
                    const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
                    const int at = cnt + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
                    if (keep) mine[at] = ed[k];
                    cnt += __popcll(m);
                }
                int tot = 0;
                for (;;) {
                    ++nround;
                    const int pb2 = rb;
                    rb = rb == 2 ? 0 : rb + 1;
                    const int cb2 = rb == 2 ? 0 : rb + 1;
                    unsigned char *bw2 = blocked[rb];
                    const unsigned char *bp2 = blocked[pb2];
                    int left = 0;
                    for (int base = 0; base < cnt; base += 64) {
                        const int idx = base + lane;
                        unsigned int e = idx < cnt ? mine[idx] : RE_NONE;
                        const int a = (int)(e >> 12), c = (int)(e & 0xfffu);
                        const int sa = state[a], ba = bp2[a];
                        bool keep = false;
                        if (e != RE_NONE && sa != 2) {
                            if (sa == 1 || ba == 0) state[c] = 2;
                            else { bw2[c] = 1; keep = true; }
                        }
                        const unsigned long long m = __builtin_amdgcn_ballot_w64(keep);
                        const int at = left + (int)__builtin_amdgcn_mbcnt_hi((unsigned int)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned int)m, 0u));
                        if (keep) mine[at] = e;           // at <= idx: behind every entry still to be read (a wave's LDS operations are in order)
                        left += __popcll(m);
                    }
                    cnt = __builtin_amdgcn_readfirstlane(left);
                    *(unsigned int *)&blocked[cb2][4 * tid] = 0u;
                    if (lane == 0) {
                        wcnt[wave] = cnt;
                        if (cnt) atomicAdd(&pend[rb], cnt);          // the flag of the rounds above, now the number of live edges
                    }
                    if (tid == 0) pend[cb2] = 0;
                    __syncthreads();
                    tot = pend[rb];
                    if (tot <= RE_TAIL) break;
                }
                if (tot > 0) {
                    // ---- the tail: at most RE_TAIL live edges.  ONE wave takes them all into registers and finishes the rounds without
                    // a barrier (a round of the whole workgroup costs ~1.6 k cycles before any edge is looked at; the chain of
                    // decisions still has 6-10 links to go); the other waves wait at the barrier in front of the emit.
                    if (wave == 0) {
                        unsigned int *comb = sedge + 16 * (RE_REG * 64);
                        int off = 0;
                        for (int w = 0; w < 16; ++w) {
                            const int n = __builtin_amdgcn_readfirstlane(wcnt[w]);
                            for (int i = lane; i < n; i += 64) comb[off + i] = sedge[w * (RE_REG * 64) + i];
                            off += n;
                        }
                        unsigned int et[RE_TAIL / 64];
#pragma unroll
                        for (int k = 0; k < RE_TAIL / 64; ++k) et[k] = lane + 64 * k < off ? comb[lane + 64 * k] : RE_NONE;
                        // blocked[rb] holds the marks of the last round (exactly the live edges' later endpoints) and the buffer cleared
                        // in that round is clean: those two alternate from here on, every round wiping the marks it consumed
                        // (offsets into the LDS array, not pointers: a swapped pointer loses its address space and every access
                        // becomes a flat one -- measured: the tail 18 us slower than no tail at all)
                        unsigned char *const bl0 = &blocked[0][0];
                        int X = rb * NMS_CAP, Y = (rb == 2 ? 0 : rb + 1) * NMS_CAP;
                        for (;;) {
                            int sa[RE_TAIL / 64], ba[RE_TAIL / 64];
#pragma unroll
                            for (int k = 0; k < RE_TAIL / 64; ++k) {
                                sa[k] = state[et[k] >> 12];
                                ba[k] = bl0[X + (int)(et[k] >> 12)];
                            }
                            bool any = false;
#pragma unroll
                            for (int k = 0; k < RE_TAIL / 64; ++k) {
                                const unsigned int e = et[k];
                                if (e == RE_NONE) continue;
                                const int c = (int)(e & 0xfffu);
                                bl0[X + c] = 0;                                      // the mark this edge set last round: consumed
                                if (sa[k] == 2) et[k] = RE_NONE;
                                else if (sa[k] == 1 || ba[k] == 0) { state[c] = 2; et[k] = RE_NONE; }
                                else { bl0[Y + c] = 1; any = true; }
                            }
                            if (__builtin_amdgcn_ballot_w64(any) == 0ull) break;
                            const int t = X; X = Y; Y = t;
                        }
                    }
                    __syncthreads();                      // the emit below reads every candidate's state
                }
                break;
            }
#endif
        }
    } else {
        // ---- fallback: sort all candidates by (score desc, anchor index asc), walk them one at a time
        unsigned long long *keys = (unsigned long long *)sedge;       // [NMS_CAP]
        for (int i = tid; i < NMS_CAP; i += 1024)
            keys[i] = i < M ? (((unsigned long long)(~__float_as_uint(cs[i])) << 32) | ((unsigned long long)(unsigned int)co[i] << 12) |
                               (unsigned int)i)
                            : ~0ull;
        __syncthreads();
        for (int k = 2; k <= NMS_CAP; k <<= 1) {
            for (int j = k >> 1; j > 0; j >>= 1) {
                for (int i = tid; i < NMS_CAP; i += 1024) {
                    const int ixj = i ^ j;
                    if (ixj > i) {
                        const unsigned long long x = keys[i], y = keys[ixj];
                        const bool asc = (i & k) == 0;
                        if ((x > y) == asc) { keys[i] = y; keys[ixj] = x; }
                    }
                }
                __syncthreads();
            }
        }
        for (int r = 0; r < M; ++r) {
            const int pi = (int)(keys[r] & 0xfffu);
            if (state[pi] == 2) continue;                           // uniform: written before the last barrier
            const float4 bi = *(const float4 *)(cb + (size_t)pi * 4);
            const float ai = (bi.z - bi.x) * (bi.w - bi.y);
            const int ci = cc[pi];
            for (int r2 = r + 1 + tid; r2 < M; r2 += 1024) {
                const int pj = (int)(keys[r2] & 0xfffu);
                if (state[pj] != 2 && cc[pj] == ci) {
                    const float4 bj = *(const float4 *)(cb + (size_t)pj * 4);
                    const float aj = (bj.z - bj.x) * (bj.w - bj.y);
                    if (suppresses_exact(bi, ai, bj, aj, thr)) state[pj] = 2;
                }
            }
            __syncthreads();
        }
        for (int pos = tid; pos < M; pos += 1024)
            if (state[pos] == 0) state[pos] = 1;
        __syncthreads();
    }
    NSTAMP(2, blockIdx.x, 2);
    // (rounds: an endpoint that nobody killed is still "undecided" here -- it is kept)
    // ---- emit: survivors in anchor-index order
    for (int pos = tid; pos < M; pos += 1024) {
        if (state[pos] != 2) {
            const int n = (int)skey[2 * pos + 1];
            atomicOr(&keepn[n >> 6], 1ull << (n & 63));
        }
    }
    __syncthreads();
    if (tid < 64) {
        const int cnt = __popcll(keepn[tid]);
        int incl = cnt;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int t = __shfl_up(incl, o, 64);
            if (lane >= o) incl += t;
        }
        wbase[tid] = incl - cnt;
        if (tid == 63) p.out_count[b] = incl < p.max_det ? incl : p.max_det;
    }
    __syncthreads();
    float *ob = p.out_box + (size_t)b * p.max_det * 4;
    float *os = p.out_score + (size_t)b * p.max_det;
    int *oc = p.out_cls + (size_t)b * p.max_det;
    for (int pos = tid; pos < M; pos += 1024) {
        if (state[pos] != 2) {
            const int n = (int)skey[2 * pos + 1];
            const unsigned long long bits = keepn[n >> 6];
            const int dst = wbase[n >> 6] + __popcll(bits & ((1ull << (n & 63)) - 1ull));
            if (dst < p.max_det) {
                *(float4 *)(ob + (size_t)dst * 4) = *(const float4 *)(cb + (size_t)pos * 4);
                os[dst] = cs[pos];
                oc[dst] = cc[pos];
            }
        }
    }
    NSTAMP(2, blockIdx.x, 7);
}

unsigned long long *y355_nms_stamps_dev = nullptr;

int y355_prepare_head(void) {
    if (int e = (int)hipFuncSetAttribute((const void *)decode_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024)) return e;
    if (int e = (int)hipFuncSetAttribute((const void *)pairs_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, PAIRS_LDS)) return e;
    return (int)hipFuncSetAttribute((const void *)pairs_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, PAIRS_LDS);
}

void y355_launch_head_nms(const HeadParams &p_in, int batch, const y355_head_ws &ws, hipStream_t s, hipEvent_t mid,
                          hipEvent_t (*kev)[2]) {
    // Heads with 3 .. MAXG classes sort their candidates (class, bin): NMS is per class (models/slim_yolo_v2.py:196-203,
    // models/tiny_yolo_v3.py:129-136), so a candidate's partners all sit in its class's bins.  The bin grid (any grid over the
    // clamped centres works) shrinks until C grids fit the CAP-entry tables.  Two-class heads keep the size groups: there the
    // class split was measured slower (profiles/r04_notes.md).
    HeadParams p = p_in;
    p.cls_groups = 0;
    if (p.C >= 3 && p.C <= MAXG) {
        int hb = p.Hb, wb = p.Wb;
        while (p.C * hb * wb > NMS_CAP && (hb > 4 || wb > 4)) {
            if (hb >= wb) --hb; else --wb;
        }
        if (p.C * hb * wb <= NMS_CAP) { p.cls_groups = 1; p.Hb = hb; p.Wb = wb; }
    }
    hipEvent_t none[2] = {nullptr, nullptr};
    hipEvent_t *k0 = kev ? kev[0] : none, *k1 = kev ? kev[1] : none, *k2 = kev ? kev[2] : none, *k3 = kev ? kev[3] : none;
    HeadWork wk;
    wk.cbox = (float *)ws.cbox;
    wk.cscore = (float *)ws.cscore;
    wk.ccls = (int *)ws.ccls;
    wk.corig = (int *)ws.corig;
    wk.count = (int *)ws.count;
    wk.edges = (unsigned int *)ws.edges;
    wk.nedges = (int *)ws.nedges;
    wk.binstart = (int *)ws.binstart;
    wk.astat = (float *)ws.astat;
    wk.tiny = (int *)ws.tiny;
    wk.ntiny = (int *)ws.ntiny;
    static unsigned long long *g_stamps = nullptr;
#ifdef Y355_EXPERIMENTS
    static const bool want = getenv("Y355_NMS_STAMPS") != nullptr;
#else
    constexpr bool want = false;
#endif
    if (want && !g_stamps) {
        if (hipMalloc((void **)&g_stamps, 8 * 8 * 256 * 4) != hipSuccess) g_stamps = nullptr;
        else (void)hipMemset(g_stamps, 0, 8 * 8 * 256 * 4);
    }
    wk.stamps = g_stamps;
    y355_nms_stamps_dev = g_stamps;
    wk.ctype = (int *)ws.ctype;
    wk.dbox = (float *)ws.dbox;
    wk.dscore = (float *)ws.dscore;
    wk.dcls = (int *)ws.dcls;
    {
        int n = 0;
        for (int l = 0; l < p.nlev; ++l) n += p.lev[l].Hs * p.lev[l].Ws * p.A;
        const bool large = n > NMS_CAP;                     // callers allocate the raw arrays for such heads
        wk.rbox = large ? (float *)ws.rbox : nullptr;
        wk.rscore = (float *)ws.rscore;
        wk.rcls = (int *)ws.rcls;
        wk.rcount = (int *)ws.rcount;
        wk.ovf = (int *)ws.ovf;
        wk.rstride = ws.rstride;
        // every level's map has the same element size and padded channel count
        // every level's map has the same element size and padded channel count; a workgroup stages dec_cells(A) cells, fewer
        // when a cell is wide (few anchors x an fp32 map of 256 channels would need 263 KB): at most DEC_LDS bytes
        constexpr int DEC_LDS = 128 * 1024;
        int cellb = 0;
        for (int l = 0; l < p.nlev; ++l) cellb = std::max(cellb, p.lev[l].cstride * (p.lev[l].pred ? 1 : 4));
        const int pitch = cellb + 4;
        const int ncell = std::max(1, std::min(dec_cells(p.A), DEC_LDS / pitch));
        int nblk = 0;
        for (int l = 0; l < p.nlev; ++l) nblk += (p.lev[l].Hs * p.lev[l].Ws + ncell - 1) / ncell;
        Y355_LAUNCH(decode_kernel, dim3(nblk, batch), dim3(DEC_THREADS), (size_t)ncell * pitch, s, k0[0], k0[1], p, wk, pitch, ncell);
        if (large) hipLaunchKernelGGL(compact_kernel, dim3(batch), dim3(1024), 0, s, p, wk);
    }
    Y355_LAUNCH(head_kernel, dim3(batch), dim3(1024), 0, s, k1[0], k1[1], p, wk);
    if (mid) (void)hipEventRecord(mid, s);
    if (p.nms_thresh >= 1e-4f && p.nms_thresh < 1e4f)
        Y355_LAUNCH(pairs_kernel<true>, dim3(p.pairs_wgs > 0 ? p.pairs_wgs : Y355_PAIRS_G, batch), dim3(1024), PAIRS_LDS, s, k2[0], k2[1], p, wk, p.nms_thresh);
    else
        Y355_LAUNCH(pairs_kernel<false>, dim3(p.pairs_wgs > 0 ? p.pairs_wgs : Y355_PAIRS_G, batch), dim3(1024), PAIRS_LDS, s, k2[0], k2[1], p, wk, p.nms_thresh);
    Y355_LAUNCH(resolve_emit_kernel, dim3(batch), dim3(1024), 0, s, k3[0], k3[1], p, wk, p.nms_thresh);
}
